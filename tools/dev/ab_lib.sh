#!/bin/bash
# A/B of two library builds on one box: tools/dev/ab_libs/lib_a.so (EX4D_HIP_LIB) against the in-tree build;  usage: ab_lib.sh [bench flags]
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
for rep in 1 2 3; do for lib in $root/tools/dev/ab_libs/lib_a.so ""; do
  EX4D_HIP_LIB=$lib timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-model-step "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stage_ms']; print('[${lib:+A}${lib:-B (in-tree)}]'[:14], d['value'], d['step_ms']['p50'], {k: round(v, 4) for k, v in s.items()})"
done; done
