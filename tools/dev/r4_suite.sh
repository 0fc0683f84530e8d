#!/bin/bash
# full GPU suite (+ optionally the forced-128-register probe libraries of tools/dev/spill_probe.py build_fix, when present)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
python -m ex4dgs_amd.build > /dev/null 2>&1
if [ -d tools/dev/spill_variants ]; then
  timeout 300 python tools/dev/spill_probe.py variants tools/dev/spill_variants > $out/r04_spill_fix.txt 2>&1
  cat $out/r04_spill_fix.txt
fi
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=1200 > $out/r04_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r04_pytest.txt
cp $out/parity_report.json $out/r04_parity_report.json 2>/dev/null
tail -4 $out/r04_pytest.txt
