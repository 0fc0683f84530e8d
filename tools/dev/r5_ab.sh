#!/bin/bash
# A/B on one box: alternating bench runs with environment settings;  usage: r5_ab.sh "ENV=.. ENV2=.." "ENV=.." [...]   (each run: --steps 40)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for rep in 1 2 3; do
  for envs in "$@"; do
    env $envs python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-model-step 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stage_ms']; print('$envs', d['value'], d['step_ms']['p50'], 'fwd', s['composite_fwd'], 'zero', s['zero_accumulators'], 'bwd', s['composite_bwd'])"
  done
done
