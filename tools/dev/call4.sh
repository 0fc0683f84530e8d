set -x
mkdir -p gpurun_out
timeout 600 python tools/dev/calib_parity.py 4 6 2>&1 | grep -v amdgpu.ids | cut -c1-150 > gpurun_out/c4_calib.txt
for v in 0 4 5 6 9; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-model-step --bwd-variant $v > gpurun_out/c4_bench_v$v.json 2> gpurun_out/c4_bench_v$v.err; done
cat gpurun_out/c4_calib.txt
python - <<'P'
import json
for v in (0,4,5,6,9):
    try:
        d=json.loads(open(f'gpurun_out/c4_bench_v{v}.json').read().strip().splitlines()[-1])
        print(v, d['value'], {k:v for k,v in d['roofline']['stage_ms'].items() if 'bwd' in k or 'composite' in k})
    except Exception as e: print(v,'ERR',e)
P
bash tools/pmc.sh c4sq_v4 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" --bwd-variant 4 | grep -i "composite"
bash tools/pmc.sh c4sq2_v4 "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS" --bwd-variant 4 | grep -i "composite"
