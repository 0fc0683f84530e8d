#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
for sr in 0 1 2 4 8; do
timeout 200 python bench.py --no-cpu-baseline --no-model-step --stripe-rows $sr > $out/r3c13_bench_sr$sr.json 2> $out/r3c13_bench_sr$sr.err
done
python - <<PY
import json
for sr in (0, 1, 2, 4, 8):
    f = "r3c13_bench_sr%d.json" % sr
    try:
        b = json.load(open("$out/" + f)); st = b["roofline"]["stage_ms"]; print(f, b["value"], b["step_ms"]["p50"], st["composite_fwd"], st["composite_bwd"])
    except Exception as e: print(f, "failed", e)
PY
_EX4D=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout=600 -k "cfg3 or sweep or subpixel" > $out/r3c13_pytest.txt 2>&1
tail -3 $out/r3c13_pytest.txt
