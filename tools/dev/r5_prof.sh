#!/bin/bash
# usage: tools/dev/r5_prof.sh <tag> "<bench args>" ["<bench args 2>" ...]  -- rocprofv3 kernel stats of short bench runs, forward kernels listed
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
i=0
for args in "$@"; do
  bash $root/tools/prof.sh ${tag}_$i $args > /dev/null 2>&1
  echo "== ${tag}_$i: $args"
  python - <<PY
import json
try:
    d=json.loads(open("$root/gpurun_out/${tag}_${i}_bench.json").read().strip().splitlines()[-1]); print("ms/frame", d["value"], d["step_ms"]["p50"])
except Exception as e: print("no bench line", e)
PY
  grep -v "at::native\|rocclr" $root/gpurun_out/${tag}_${i}_kernel_stats.txt | cut -c1-62,71-120 | head -24
  i=$((i+1))
done
