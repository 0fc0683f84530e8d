#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m ex4dgs_amd.build > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -q -p no:cacheprovider > $out/r04f_pytest_round4.txt 2>&1
echo "pytest rc=$?" >> $out/r04f_pytest_round4.txt
tail -25 $out/r04f_pytest_round4.txt
