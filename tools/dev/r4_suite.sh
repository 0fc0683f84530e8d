#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m ex4dgs_amd.build > /dev/null 2>&1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=1200 > $out/r04_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r04_pytest.txt
cp $out/parity_report.json $out/r04_parity_report.json 2>/dev/null
tail -8 $out/r04_pytest.txt
