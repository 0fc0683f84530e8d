#!/bin/bash
# A/B of one library option on one box;  usage: ab_opt.sh <option> <value a> <value b> [bench flags]
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
opt=$1; va=$2; vb=$3; shift; shift; shift
for rep in 1 2 3; do for v in $va $vb; do
  timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-model-step --set $opt=$v "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stage_ms']; print('[$opt=$v]', d['value'], d['step_ms']['p50'], {k: round(v, 4) for k, v in s.items()})"
done; done
