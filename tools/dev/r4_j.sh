#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m ex4dgs_amd.build > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py -m gpu -q -x -p no:cacheprovider -k "cfg1 or zero_dir3D or static_20k or dynamic_keyframed or degenerate or tile or huge or asynchronous or graph or deep_overlap or overflow or bench" > $out/r04j_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r04j_pytest.txt
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-model-step"
for i in 1 2 3; do
timeout 300 $B 2>> $out/r04j_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg3: ms/frame', d['value'], 'p50', d['step_ms']['p50'], json.dumps(d['roofline']['stage_ms']))" >> $out/r04j_modes.txt 2>&1
done
tail -5 $out/r04j_pytest.txt; cat $out/r04j_modes.txt
