#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > $out/r03_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r03_pytest.txt
cp $out/parity_report.json $out/r03_parity_report.json 2>/dev/null
timeout 1500 bash tools/prof_round.sh r03 > $out/r03_prof_script.log 2>&1
tail -5 $out/r03_pytest.txt
cat $out/r03_bench_default.json | cut -c1-1500
tail -30 $out/r03_prof_script.log | cut -c1-170
