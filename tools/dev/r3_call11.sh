#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > $out/r3c11_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r3c11_pytest.txt
cp $out/parity_report.json $out/r3c11_parity_report.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --no-model-step > $out/r3c11_bench.json 2> $out/r3c11_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-model-step > $out/r3c11_prof.log 2>&1
python $root/tools/rocpd_summary.py $(find /tmp/prof_b -name "*.db" | head -1) $out/r3c11_kernel_stats.txt
cd $root
tail -6 $out/r3c11_pytest.txt
python - <<PY
import json
for f in ("r3c11_bench.json",):
    try:
        b = json.load(open("$out/" + f)); print(f, b["value"], b["step_ms"]["p50"], b["roofline"]["stage_ms"])
    except Exception as e: print(f, "failed", e)
PY
head -12 $out/r3c11_kernel_stats.txt | cut -c1-140
