#!/bin/bash
# round 4: the round's measurement artefacts (kernel-trace stats + PMC passes of the default bench command, the fused training iteration,
# the compiled trainer, distCUDA2) -> gpurun_out/r04_*, profiles/r04_pmc_traffic.json, and the default bench line that quotes them
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m ex4dgs_amd.build > /dev/null 2>&1
timeout 1700 bash tools/prof_round.sh r04 > $out/r04_prof_script.log 2>&1
tail -45 $out/r04_prof_script.log | cut -c1-170
cut -c1-2500 $out/r04_bench_default.json
