#!/bin/bash
# usage (GPU box, repo root): tools/pmc.sh <tag> "<counter list>" [bench args]   -> gpurun_out/<tag>_pmc.db
tag=$1; ctrs=$2; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_$tag -o $tag -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-model-step "$@" > $root/gpurun_out/${tag}_pmc.log 2>&1
db=$(find /tmp/pmc_$tag -name "*.db" | head -1)
python $root/tools/pmc_summary.py $db > $root/gpurun_out/${tag}_pmc.txt 2>&1
cat $root/gpurun_out/${tag}_pmc.txt | cut -c1-60,161-400 | head -40
