set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -30 > gpurun_out/c6_dist.txt
timeout 600 python bench.py > gpurun_out/c6_bench_default.json 2> gpurun_out/c6_bench_default.err
timeout 600 python bench.py --config cfg4 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c6_bench_cfg4.json 2> gpurun_out/c6_bench_cfg4.err
timeout 600 python bench.py --train-core --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c6_bench_cfg3_train.json 2> gpurun_out/c6_bench_cfg3_train.err
timeout 600 python bench.py --gpus 2 --share-device --backend gloo --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/c6_bench_2r.json 2> gpurun_out/c6_bench_2r.err
cat gpurun_out/c6_dist.txt; tail -3 gpurun_out/*.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/c6_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['step_ms'], d.get('multi_gpu'), d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['stage_ms'])
        for k in ('cpu_baseline','cpu_oracle_cfg2_full','cpu_torch_baseline','model_step'): 
            if k in d: print('   ',k, str(d[k])[:300])
    except Exception as e: print(f,'ERR',e)
P
