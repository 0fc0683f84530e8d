#!/bin/bash
# round 5, call A: the MSD depth sort -- tests, then A/B bench lines
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m ex4dgs_amd.build > $out/r05a_build.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_round5.py -q -p no:cacheprovider --timeout=600 -x > $out/r05a_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r05a_pytest.txt
tail -15 $out/r05a_pytest.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider --timeout=600 -x -k "depth_ties or odd_sizes or empty_and or cfg1_forward or dynamic_keyframed or technicolor or 65535 or tile_sort_with" > $out/r05a_pytest2.txt 2>&1
echo "pytest rc=$?" >> $out/r05a_pytest2.txt
tail -5 $out/r05a_pytest2.txt
timeout 600 python -m pytest tests/test_gpu_round4.py -q -p no:cacheprovider --timeout=600 -x > $out/r05a_pytest3.txt 2>&1
echo "pytest rc=$?" >> $out/r05a_pytest3.txt
tail -5 $out/r05a_pytest3.txt
for m in 0 1 2 0 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-model-step --set depth_sort_msd=$m > $out/r05a_bench_msd$m.json 2>$out/r05a_bench_msd$m.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/r05a_bench_msd$m.json").read().strip().splitlines()[-1])
    print("msd=$m", d["value"], d["step_ms"], d["roofline"]["stage_ms"])
except Exception as e:
    print("msd=$m failed", e); print(open("$out/r05a_bench_msd$m.err").read()[-2000:])
PY
done
