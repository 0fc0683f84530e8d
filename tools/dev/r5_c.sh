#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
for args in "--set depth_sort_msd=2" "--set depth_sort_local_cap=268435456" "--set depth_sort_local_cap=536870912" "--set depth_sort_local_cap=1073741824" "--set depth_sort_local_cap=1610612736"; do
  bash $root/tools/prof.sh r05d_$i $args > /dev/null 2>&1
  echo "== $args"; grep "depth_local\|rs_scatter_kernel<8" $root/gpurun_out/r05d_${i}_kernel_stats.txt | cut -c1-62,71-120
  i=$((i+1))
done
