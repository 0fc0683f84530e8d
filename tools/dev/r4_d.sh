#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m ex4dgs_amd.build > /dev/null 2>&1
timeout 300 python tools/dev/dev_loss_time.py > $out/r04d_loss_time.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_loss; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_loss -o loss -- python $root/tools/dev/dev_loss_time.py > /dev/null 2>&1
python $root/tools/rocpd_summary.py $(find /tmp/prof_loss -name "*.db" | head -1) $out/r04d_loss_kernel_stats.txt > /dev/null 2>&1
cd $root
tail -3 $out/r04d_loss_time.txt; grep "l1_ssim" $out/r04d_loss_kernel_stats.txt | cut -c1-130
