#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout 300 python tools/dev/dbg_fuzz_case.py 10 7 0.0 3 > $out/r03_dbg_case10.txt 2>&1
tail -40 $out/r03_dbg_case10.txt | cut -c1-400
