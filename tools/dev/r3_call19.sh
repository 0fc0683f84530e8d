#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout=600 -k "cfg1 or small or cfg3 or subpixel or smoke" > $out/r3c19_pytest_a.txt 2>&1
tail -4 $out/r3c19_pytest_a.txt
for o in 0 1 0 1; do
timeout 200 python bench.py --no-cpu-baseline --no-model-step --fwd-asm $o > $out/r3c19_bench_$o.json 2> $out/r3c19_bench_$o.err
python - <<PY
import json
f = "r3c19_bench_$o.json"
try:
    b = json.load(open("$out/" + f)); st = b["roofline"]["stage_ms"]; print(f, b["value"], b["step_ms"]["p50"], st["composite_fwd"], st["composite_bwd"])
except Exception as e: print(f, "failed", e); print(open("$out/r3c19_bench_$o.err").read()[-500:])
PY
done
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout=600 > $out/r3c19_pytest.txt 2>&1
tail -4 $out/r3c19_pytest.txt
