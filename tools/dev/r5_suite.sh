#!/bin/bash
# full GPU suite -> gpurun_out/r05_pytest.txt (+ parity report)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m ex4dgs_amd.build > /dev/null 2>&1
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=1200 "$@" > $out/r05_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r05_pytest.txt
cp $out/parity_report.json $out/r05_parity_report.json 2>/dev/null
tail -15 $out/r05_pytest.txt
