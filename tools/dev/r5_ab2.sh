#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for rep in 1 2; do
 for cfg in cfg3 cfg2; do
  for flag in "" "--sync-forward"; do
    python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-model-step --train-core --config $cfg $flag 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg train-core $flag', d['value'], d['step_ms']['p50'])"
  done
 done
done
