#!/bin/bash
# round 3, GPU call 3: forward-kernel experiment (variants 0 coop, 1 quad/64/no rows, 2 quad/64/rows, 3 quad/32/rows, 4 quad/32/no rows)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q -x -p no:cacheprovider --timeout=600 -k "not two_ranks" > $out/r3c3_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r3c3_pytest.txt
tail -4 $out/r3c3_pytest.txt
for v in 0 1 2 3 4; do
  timeout 120 python bench.py --no-cpu-baseline --no-model-step --steps 30 --fwd-variant $v > $out/r3c3_bench_v$v.json 2> $out/r3c3_bench_v$v.err
  python - <<PY
import json
try:
    b = json.load(open("$out/r3c3_bench_v$v.json"))
    print("variant $v", b["value"], {k: b["roofline"]["stage_ms"][k] for k in ("composite_fwd", "composite_bwd")})
except Exception as e:
    print("variant $v failed", e)
PY
done
for v in 0 1 2 3; do
  bash tools/pmc.sh r3c3_pmcA_v$v "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" --fwd-variant $v 2>&1 | grep -i "composite_fwd" | cut -c1-260
done
