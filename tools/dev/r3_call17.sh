#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
for o in 2 3 0 2 3; do
timeout 200 python bench.py --no-cpu-baseline --no-model-step --fwd-variant $o > $out/r3c17_bench_v$o.json 2> $out/r3c17_bench_v$o.err
python - <<PY
import json
f = "r3c17_bench_v$o.json"
try:
    b = json.load(open("$out/" + f)); st = b["roofline"]["stage_ms"]; print(f, b["value"], b["step_ms"]["p50"], st["composite_fwd"], st["composite_bwd"])
except Exception as e: print(f, "failed", e)
PY
done
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cfg3 or sweep or subpixel or flow or small" -p no:cacheprovider --timeout=600 > $out/r3c17_pytest.txt 2>&1
tail -3 $out/r3c17_pytest.txt
