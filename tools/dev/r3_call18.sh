#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
for i in 1 2; do
timeout 200 python bench.py --no-cpu-baseline --no-model-step > $out/r3c18_bench_$i.json 2> $out/r3c18_bench_$i.err
python - <<PY
import json
f = "r3c18_bench_$i.json"
try:
    b = json.load(open("$out/" + f)); st = b["roofline"]["stage_ms"]; print(f, b["value"], b["step_ms"]["p50"], st["preprocess_fwd"], st["preprocess_bwd"], st["composite_fwd"], st["composite_bwd"])
except Exception as e: print(f, "failed", e)
PY
done
timeout 500 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout=600 -k "prepared or trainer or native or cfg3 or sweep or autograd or wrapper" > $out/r3c18_pytest.txt 2>&1
tail -3 $out/r3c18_pytest.txt
