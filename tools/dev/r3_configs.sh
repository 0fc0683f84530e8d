#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 300 python bench.py --config cfg2 --no-cpu-baseline --no-model-step > $out/r03_bench_cfg2.json 2> $out/r03_bench_cfg2.err
timeout 400 python bench.py --config cfg4 --no-cpu-baseline --no-model-step --steps 30 > $out/r03_bench_cfg4.json 2> $out/r03_bench_cfg4.err
timeout 400 python bench.py --config cfg4 --no-cpu-baseline --no-model-step --steps 30 --optimizer none > $out/r03_bench_cfg4_no_optimizer.json 2> $out/r03_bench_cfg4_no_optimizer.err
timeout 300 python bench.py --config cfg5 --forward-only --no-cpu-baseline --no-model-step > $out/r03_bench_cfg5_forward_only.json 2> $out/r03_bench_cfg5_forward_only.err
timeout 300 python bench.py --train-core --no-cpu-baseline --no-model-step > $out/r03_bench_cfg3_train_core.json 2> $out/r03_bench_cfg3_train_core.err
timeout 400 python bench.py --gpus 2 --share-device --backend gloo --steps 10 --warmup 3 --no-cpu-baseline --no-model-step > $out/r03_bench_2ranks_one_gpu_gloo.json 2> $out/r03_bench_2ranks_one_gpu_gloo.err
for f in cfg2 cfg4 cfg4_no_optimizer cfg5_forward_only cfg3_train_core 2ranks_one_gpu_gloo; do
python - <<PY
import json
try:
    b = json.load(open("$out/r03_bench_$f.json")); print("$f", b["value"], b["unit"], b.get("step_ms", {}).get("p50"), json.dumps(b.get("multi_gpu"))[:400])
except Exception as e:
    print("$f failed", e); print(open("$out/r03_bench_$f.err").read()[-600:])
PY
done
