#!/bin/bash
# several values of one option on one box;  usage: ab_multi.sh <option> <v1> <v2> ...
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
opt=$1; shift
for rep in 1 2; do for v in "$@"; do
  timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-model-step --set $opt=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stage_ms']; print('[$opt=$v]', d['value'], d['step_ms']['p50'], 'preprocess_fwd', s['preprocess_fwd'])"
done; done
