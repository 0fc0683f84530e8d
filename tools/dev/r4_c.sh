#!/bin/bash
# round 4, third GPU call: one-wave workgroups / priorities for preprocess_fwd, rolling-window loss kernels (tests + time + counters), scratch probe
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m ex4dgs_amd.build > /dev/null 2>&1
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-model-step"
for tune in 0 4 5 7 6 0 5; do
  EX4D_PREPROCESS_TUNE=$tune timeout 300 $B 2> $out/r04c_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('tune $tune: ms/frame', d['value'], 'stages', json.dumps(d['roofline']['stage_ms']))" >> $out/r04c_experiments.txt 2>&1
done
EX4D_PREPROCESS_TUNE=7 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "cfg1 or zero_dir3D or sh_degrees or static_20k or dynamic_keyframed or split or degenerate" > $out/r04c_pytest_tune7.txt 2>&1
echo "pytest rc=$?" >> $out/r04c_pytest_tune7.txt
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "loss or edge_cases or native or trainer" > $out/r04c_pytest_loss.txt 2>&1
echo "pytest rc=$?" >> $out/r04c_pytest_loss.txt
timeout 300 python tools/dev/dev_loss_time.py > $out/r04c_loss_time.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_loss_$ctr
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_loss_$ctr -o loss -- python $root/tools/dev/dev_loss_time.py > $out/r04c_loss_pmc_$ctr.log 2>&1
  python $root/tools/pmc_summary.py $(find /tmp/pmc_loss_$ctr -name "*.db" | head -1) 2>&1 | grep "kernel \|l1_ssim" > $out/r04c_loss_$ctr.txt
done
rm -rf /tmp/prof_loss; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_loss -o loss -- python $root/tools/dev/dev_loss_time.py > /dev/null 2>&1
python $root/tools/rocpd_summary.py $(find /tmp/prof_loss -name "*.db" | head -1) $out/r04c_loss_kernel_stats.txt > /dev/null 2>&1
cd $root
hipcc -O3 --offload-arch=gfx950 tools/dev/micro/scratch_probe.hip -o /tmp/scratch_probe 2> /dev/null && timeout 120 /tmp/scratch_probe > $out/r04c_scratch_probe.txt 2>&1
cat $out/r04c_experiments.txt
tail -3 $out/r04c_pytest_tune7.txt; tail -5 $out/r04c_pytest_loss.txt
cat $out/r04c_loss_time.txt | tail -3; cat $out/r04c_loss_FETCH_SIZE.txt $out/r04c_loss_WRITE_SIZE.txt | cut -c1-110; grep "l1_ssim" $out/r04c_loss_kernel_stats.txt | cut -c1-120
cat $out/r04c_scratch_probe.txt
