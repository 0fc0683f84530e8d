#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
for o in 0 1 0 1; do
timeout 200 python bench.py --no-cpu-baseline --no-model-step --tile-order $o > $out/r3c14_bench_o$o.json 2> $out/r3c14_bench_o$o.err
python - <<PY
import json
f = "r3c14_bench_o$o.json"
try:
    b = json.load(open("$out/" + f)); st = b["roofline"]["stage_ms"]; print(f, b["value"], b["step_ms"]["p50"], st["composite_fwd"], st["composite_bwd"])
except Exception as e: print(f, "failed", e)
PY
done
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q -x -p no:cacheprovider --timeout=600 -k "cfg3 or sweep or subpixel or qlist or second_backward" > $out/r3c14_pytest.txt 2>&1
tail -3 $out/r3c14_pytest.txt
