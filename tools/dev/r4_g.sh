#!/bin/bash
# full -m gpu suite + default bench + cfg2 modes
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m ex4dgs_amd.build > /dev/null 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=1200 > $out/r04g_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r04g_pytest.txt
cp $out/parity_report.json $out/r04g_parity_report.json 2>/dev/null
B="python bench.py --no-cpu-baseline --no-model-step"
for mode in "" "--async-frames" "--graph"; do
  timeout 300 $B --config cfg2 --steps 100 --warmup 20 $mode 2>> $out/r04g_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg2 [$mode]: ms/frame', d['value'], 'p50', d['step_ms']['p50'], '|', d['config']['step'][-90:], json.dumps(d['roofline']['stage_ms']))" >> $out/r04g_modes.txt 2>&1
done
timeout 300 $B --steps 50 --warmup 10 2>> $out/r04g_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg3: ms/frame', d['value'], 'p50', d['step_ms']['p50'], json.dumps(d['roofline']['stage_ms']))" >> $out/r04g_modes.txt 2>&1
tail -12 $out/r04g_pytest.txt
cat $out/r04g_modes.txt
