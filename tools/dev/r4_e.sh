#!/bin/bash
# round 4, GPU call: asynchronous forward (tests), cfg2 sync / async / graph bench lines, native trainer timings
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m ex4dgs_amd.build > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -q -x -p no:cacheprovider > $out/r04e_pytest_round4.txt 2>&1
echo "pytest rc=$?" >> $out/r04e_pytest_round4.txt
B="python bench.py --no-cpu-baseline --no-model-step"
for mode in "" "--async-frames" "--graph" "" "--async-frames" "--graph"; do
  timeout 300 $B --config cfg2 --steps 100 --warmup 20 $mode 2>> $out/r04e_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg2 [$mode]: ms/frame', d['value'], 'p50', d['step_ms']['p50'], '|', d['config']['step'][-90:])" >> $out/r04e_modes.txt 2>&1
done
for mode in "" "--async-frames" "--graph"; do
  timeout 300 $B --steps 50 --warmup 10 $mode 2>> $out/r04e_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg3 [$mode]: ms/frame', d['value'], 'p50', d['step_ms']['p50'])" >> $out/r04e_modes.txt 2>&1
done
timeout 300 python tools/dev/native_overhead.py 300000 60 >> $out/r04e_modes.txt 2>&1
timeout 300 python tools/dev/native_overhead.py 300000 60 async >> $out/r04e_modes.txt 2>&1
tail -15 $out/r04e_pytest_round4.txt
cat $out/r04e_modes.txt
tail -5 $out/r04e_err.txt
