#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py -q -p no:cacheprovider --timeout=600 -x > $out/r05b_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r05b_pytest.txt
tail -8 $out/r05b_pytest.txt
bash tools/dev/r5_prof.sh r05c "$@"
