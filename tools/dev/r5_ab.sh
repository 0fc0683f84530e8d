#!/bin/bash
# A/B on one box: alternating bench runs;  usage: r5_ab.sh "<env assignments and/or bench flags>" "<...>" [...]   (each run: --steps 40, 3 rounds)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for rep in 1 2 3; do
  for spec in "$@"; do
    envs=""; flags=""
    for w in $spec; do if [[ "$w" =~ ^[A-Z][A-Z0-9_]*= ]]; then envs="$envs $w"; else flags="$flags $w"; fi; done
    env $envs python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-model-step $flags 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stage_ms']; print('$spec', d['value'], d['step_ms']['p50'], {k: round(v, 4) for k, v in s.items() if k in ('depth_sort', 'scan_tiles', 'duplicate', 'tile_sort')})"
  done
done
