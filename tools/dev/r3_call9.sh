#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q -x -p no:cacheprovider --timeout=600 -k "not two_ranks" > $out/r3c9_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r3c9_pytest.txt
tail -3 $out/r3c9_pytest.txt
for cfg in "1 1" "1 2" "1 3" "1 4" "0 2" "inline"; do
  set -- $cfg
  if [ "$1" = "inline" ]; then extra="--no-side-stream"; tag=inline; else extra="--color-fork $1 --color-wgs $2"; tag=f$1w$2; fi
  timeout 120 python bench.py --no-cpu-baseline --no-model-step --steps 40 $extra > $out/r3c9_bench_$tag.json 2> $out/r3c9_bench_$tag.err
  python - <<PY
import json
try:
    b = json.load(open("$out/r3c9_bench_$tag.json")); st = b["roofline"]["stage_ms"]
    print("$tag", b["value"], "p50", b["step_ms"]["p50"], {k: st[k] for k in ("preprocess_fwd", "depth_sort", "scan_tiles", "duplicate", "tile_sort", "composite_fwd", "preprocess_bwd")})
except Exception as e: print("$tag failed", e)
PY
done
