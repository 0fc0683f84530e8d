#!/bin/bash
# round 4, second GPU call: experiments behind options, each measured by the stage timers of the default bench command
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m ex4dgs_amd.build > /dev/null 2>&1
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-model-step"
for tune in 0 1 2 3 0 3; do
  EX4D_PREPROCESS_TUNE=$tune timeout 300 $B 2> $out/r04b_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('tune $tune: ms/frame', d['value'], 'stages', json.dumps(d['roofline']['stage_ms']))" >> $out/r04b_experiments.txt 2>&1
done
# parity of the predicated SH loads / staggered priorities: a slice of the suite with the option forced on
EX4D_PREPROCESS_TUNE=3 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "cfg1 or zero_dir3D or sh_degrees or static_20k or dynamic_keyframed or split" > $out/r04b_pytest_tune3.txt 2>&1
echo "pytest rc=$?" >> $out/r04b_pytest_tune3.txt
timeout 400 python tools/dev/bwd_stats.py > $out/r04b_bwd_stats.txt 2>&1
timeout 600 python tools/dev/spill_probe.py > $out/r04b_spill_probe.txt 2>&1
cat $out/r04b_experiments.txt
tail -4 $out/r04b_pytest_tune3.txt
cat $out/r04b_bwd_stats.txt | tail -12
tail -40 $out/r04b_spill_probe.txt
