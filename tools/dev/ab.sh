#!/bin/bash
# A/B on one box: alternating bench runs;  usage: ab.sh "<bench flags>" "<bench flags>" ...   (3 rounds, stage timers of the compositing kernels and the frame)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
for rep in 1 2 3; do for spec in "$@"; do
  timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-model-step $spec 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stage_ms']; print('[$spec]', d['value'], d['step_ms']['p50'], {k: round(v, 4) for k, v in s.items()})"
done; done
