#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_round2.py -m gpu -q -x -p no:cacheprovider --timeout=600 > $out/r03_check_pytest.txt 2>&1
tail -4 $out/r03_check_pytest.txt
