#!/bin/bash
# usage (GPU box, repo root): tools/dev/quick_stats.sh <tag>  -- kernel-trace stats of a short forward+backward bench run -> gpurun_out/<tag>_kernel_stats.txt
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_q
rocprofv3 --kernel-trace --stats -d /tmp/prof_q -o q -- python $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-model-step > $out/${tag}_q.log 2>&1
python $root/tools/rocpd_summary.py $(find /tmp/prof_q -name "*.db" | head -1) $out/${tag}_kernel_stats.txt
head -24 $out/${tag}_kernel_stats.txt | cut -c1-160
