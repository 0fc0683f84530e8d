#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout=600 > $out/r3c12_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r3c12_pytest.txt
timeout 200 python bench.py --no-cpu-baseline --no-model-step > $out/r3c12_bench.json 2> $out/r3c12_bench.err
cd /tmp && export TMPDIR=/tmp
BENCH="python $root/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-model-step"
run_pmc () {     # name, counters, command...
  name=$1; ctrs=$2; shift; shift
  rm -rf /tmp/pmc_$name
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_$name -o $name -- "$@" > $out/r3c12_${name}_pmc.log 2>&1
  python $root/tools/pmc_summary.py $(find /tmp/pmc_$name -name "*.db" | head -1) > $out/r3c12_${name}.txt 2>&1
}
run_pmc SQ_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" $BENCH
run_pmc SQ_lds "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" $BENCH
run_pmc SQ_misc "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC" $BENCH
cd $root
tail -4 $out/r3c12_pytest.txt
python - <<PY
import json
for f in ("r3c12_bench.json",):
    try:
        b = json.load(open("$out/" + f)); print(f, b["value"], b["step_ms"]["p50"], b["roofline"]["stage_ms"])
    except Exception as e: print(f, "failed", e)
PY
for f in SQ_valu SQ_lds SQ_misc; do head -4 $out/r3c12_$f.txt | cut -c1-60,65-400; done
