#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args...]
# rocprofv3 kernel trace of a short bench run -> gpurun_out/<tag>_kernel_stats.txt
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag -- python $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-model-step "$@" > $root/gpurun_out/${tag}_bench.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python $root/tools/rocpd_summary.py $db $root/gpurun_out/${tag}_kernel_stats.txt
grep '"metric"' $root/gpurun_out/${tag}_bench.log > $root/gpurun_out/${tag}_bench.json
head -25 $root/gpurun_out/${tag}_kernel_stats.txt | cut -c1-70,161-250
