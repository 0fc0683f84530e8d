#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 400 python tools/dev/fuzz_parity.py 70 7 0.0 > $out/r03_fuzz_no_flow.txt 2>&1
echo "rc=$?" >> $out/r03_fuzz_no_flow.txt
tail -4 $out/r03_fuzz_no_flow.txt
grep -c " ok" $out/r03_fuzz_no_flow.txt
