set -x
mkdir -p gpurun_out
./tools/dev/micro/mfma_dpp_probe.bin > gpurun_out/c1_probe.txt 2>&1
timeout 900 python tools/dev/calib_parity.py 0 2 1 > gpurun_out/c1_calib.txt 2>&1
for v in 0 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-model-step --bwd-variant $v > gpurun_out/c1_bench_v$v.json 2> gpurun_out/c1_bench_v$v.err; done
tail -5 gpurun_out/c1_probe.txt; tail -30 gpurun_out/c1_calib.txt
python - <<'P'
import json
for v in (0,1,2):
    try:
        d=json.loads(open(f'gpurun_out/c1_bench_v{v}.json').read().strip().splitlines()[-1])
        print(v, d['value'], d['roofline']['stage_ms'])
    except Exception as e: print(v,'ERR',e)
P
