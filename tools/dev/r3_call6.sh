#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > $out/r3c6_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r3c6_pytest.txt
cp $out/parity_report.json $out/r3c6_parity_report.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-model-step > $out/r3c6_bench.json 2> $out/r3c6_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-model-step > $out/r3c6_prof.log 2>&1
python $root/tools/rocpd_summary.py $(find /tmp/prof_b -name "*.db" | head -1) $out/r3c6_kernel_stats.txt
cd $root
tail -8 $out/r3c6_pytest.txt
cut -c1-300 $out/r3c6_bench.json
head -8 $out/r3c6_kernel_stats.txt | cut -c1-140
