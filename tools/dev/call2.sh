set -x
mkdir -p gpurun_out
timeout 600 python tools/dev/calib_parity.py 2 2>&1 | grep -v amdgpu.ids | cut -c1-150 > gpurun_out/c2_calib.txt
for v in 0 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-model-step --bwd-variant $v > gpurun_out/c2_bench_v$v.json 2> gpurun_out/c2_bench_v$v.err; done
cat gpurun_out/c2_calib.txt
python - <<'P'
import json
for v in (0,1,2):
    try:
        d=json.loads(open(f'gpurun_out/c2_bench_v{v}.json').read().strip().splitlines()[-1])
        print(v, d['value'], {k:v for k,v in d['roofline']['stage_ms'].items() if 'bwd' in k or 'zero' in k})
    except Exception as e: print(v,'ERR',e)
P
