#!/bin/bash
# a subset of the GPU suite (arguments: pytest selection), then rocprofv3 kernel stats of short bench runs ("--" separates the bench argument sets)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
sel=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do sel+=("$1"); shift; done; [ "$1" == "--" ] && shift
if [ ${#sel[@]} -gt 0 ]; then
  timeout 1500 python -m pytest "${sel[@]}" -q -p no:cacheprovider --timeout=900 -x > $out/r05b_pytest.txt 2>&1
  echo "pytest rc=$?" >> $out/r05b_pytest.txt
  tail -12 $out/r05b_pytest.txt
fi
[ $# -gt 0 ] && bash tools/dev/r5_prof.sh r05c "$@"
