#!/bin/bash
# round 4, first GPU call: the whole -m gpu suite (exact dominant index, modelled end-to-end bar, 70-case fuzz, RCCL one-rank test) + the default bench line
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=1200 > $out/r04a_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r04a_pytest.txt
cp $out/parity_report.json $out/r04a_parity_report.json 2>/dev/null
timeout 600 python bench.py > $out/r04a_bench_default.json 2> $out/r04a_bench_default.err
tail -25 $out/r04a_pytest.txt
cut -c1-1800 $out/r04a_bench_default.json
