#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 200 python tools/dev/soak.py 3000 > $out/r03_soak.txt 2>&1
echo "rc=$?" >> $out/r03_soak.txt
tail -12 $out/r03_soak.txt
