#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
for o in 1 2; do
timeout 200 python bench.py --no-cpu-baseline --no-model-step > $out/r3c21_bench_$o.json 2> $out/r3c21_bench_$o.err
python - <<PY
import json
f = "r3c21_bench_$o.json"
try:
    b = json.load(open("$out/" + f)); st = b["roofline"]["stage_ms"]; print(f, b["value"], b["step_ms"]["p50"], st["composite_fwd"], st["composite_bwd"])
except Exception as e: print(f, "failed", e); print(open("$out/r3c21_bench_$o.err").read()[-500:])
PY
done
timeout 200 python bench.py --no-cpu-baseline --no-model-step --config cfg2 > $out/r3c21_bench_cfg2.json 2> /dev/null
python -c "
import json; b=json.load(open('$out/r3c21_bench_cfg2.json')); print('cfg2', b['value'], b['roofline']['stage_ms']['composite_bwd'])"
timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q -x -p no:cacheprovider --timeout=600 > $out/r3c21_pytest.txt 2>&1
tail -4 $out/r3c21_pytest.txt
