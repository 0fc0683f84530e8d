#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py -q -p no:cacheprovider --timeout=600 -x > $out/r05b_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r05b_pytest.txt
tail -12 $out/r05b_pytest.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider --timeout=600 -x -k "depth_ties or odd_sizes or empty_and or cfg1_forward or dynamic_keyframed or technicolor or 65535 or tile_sort_with or deep_overlap or full_size" > $out/r05b_pytest2.txt 2>&1
echo "pytest rc=$?" >> $out/r05b_pytest2.txt
tail -5 $out/r05b_pytest2.txt
bash tools/dev/r5_prof.sh r05c "$@"
