#!/bin/bash
# round 4, closing GPU call: the whole -m gpu suite on the final sources, then tools/prof_round.sh (kernel stats + PMC passes + the default
# bench line that quotes them)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m ex4dgs_amd.build > /dev/null 2>&1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=1200 > $out/r04_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r04_pytest.txt
cp $out/parity_report.json $out/r04_parity_report.json 2>/dev/null
timeout 1700 bash tools/prof_round.sh r04 > $out/r04_prof_script.log 2>&1
tail -6 $out/r04_pytest.txt
grep "^frame:" $out/r04_prof_script.log
cut -c1-900 $out/r04_bench_default.json
