#!/bin/bash
# usage (GPU box): tools/dev/round_configs.sh <tag> -- one bench line per other BASELINE configuration, the graph / async modes of config 2,
# the two-rank self-spawn on one GPU, and the 3000-iteration soak run -> gpurun_out/<tag>_*
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
B="--no-cpu-baseline --no-model-step"
timeout 300 python bench.py --config cfg2 $B > $out/${tag}_bench_cfg2.json 2> $out/${tag}_bench_cfg2.err
timeout 300 python bench.py --config cfg2 $B --graph > $out/${tag}_bench_cfg2_graph.json 2> $out/${tag}_bench_cfg2_graph.err
timeout 300 python bench.py $B --graph > $out/${tag}_bench_cfg3_graph.json 2> $out/${tag}_bench_cfg3_graph.err
timeout 400 python bench.py --config cfg4 $B --steps 30 > $out/${tag}_bench_cfg4.json 2> $out/${tag}_bench_cfg4.err
timeout 400 python bench.py --config cfg4 $B --steps 30 --optimizer none > $out/${tag}_bench_cfg4_no_optimizer.json 2> $out/${tag}_bench_cfg4_no_optimizer.err
timeout 300 python bench.py --config cfg5 --forward-only $B > $out/${tag}_bench_cfg5_forward_only.json 2> $out/${tag}_bench_cfg5_forward_only.err
timeout 300 python bench.py --train-core $B > $out/${tag}_bench_cfg3_train_core.json 2> $out/${tag}_bench_cfg3_train_core.err
timeout 300 python bench.py --train-core --optimizer sharded $B > $out/${tag}_bench_cfg3_train_core_sharded.json 2> $out/${tag}_bench_cfg3_train_core_sharded.err
timeout 300 python bench.py $B --set depth_sort_msd=0 > $out/${tag}_bench_cfg3_lsd_depth_sort.json 2> $out/${tag}_bench_cfg3_lsd_depth_sort.err
timeout 300 python bench.py $B --set depth_sort_msd=2 --graph > $out/${tag}_bench_cfg3_msd_depth_sort_graph.json 2> $out/${tag}_bench_cfg3_msd_depth_sort_graph.err
timeout 300 python bench.py --config cfg2 $B --async-frames > $out/${tag}_bench_cfg2_async.json 2> $out/${tag}_bench_cfg2_async.err
timeout 400 python bench.py --gpus 2 --share-device --backend gloo --steps 10 --warmup 3 $B > $out/${tag}_bench_2ranks_one_gpu_gloo.json 2> $out/${tag}_bench_2ranks_one_gpu_gloo.err
for f in cfg2 cfg2_graph cfg2_async cfg3_graph cfg3_lsd_depth_sort cfg3_msd_depth_sort_graph cfg4 cfg4_no_optimizer cfg5_forward_only cfg3_train_core cfg3_train_core_sharded 2ranks_one_gpu_gloo; do
python - <<PY
import json
try:
    b = json.load(open("$out/${tag}_bench_$f.json")); print("$f", b["value"], b["unit"], b.get("step_ms", {}).get("p50"), json.dumps(b.get("multi_gpu"))[:300])
except Exception as e:
    print("$f failed", e); print(open("$out/${tag}_bench_$f.err").read()[-600:])
PY
done
timeout 300 python tools/dev/soak.py 3000 > $out/${tag}_soak.txt 2>&1
echo "rc=$?" >> $out/${tag}_soak.txt
tail -6 $out/${tag}_soak.txt
