#!/bin/bash
# usage (GPU box, repo root): tools/prof_round.sh <tag>     (e.g. r04)
# Every measurement artefact of the round in one go -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/):
#   kernel-trace stats of the default bench command; separate --pmc passes (FETCH_SIZE, WRITE_SIZE, two SQ groups) of the same command;
#   kernel-trace stats + FETCH/WRITE passes of the fused training iteration (attributes, loss, RAdam kernels) and of distCUDA2
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run_stats () {   # name, command...
  name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- "$@" > $out/${tag}_${name}.log 2>&1
  python $root/tools/rocpd_summary.py $(find /tmp/prof_$name -name "*.db" | head -1) $out/${tag}_${name}_kernel_stats.txt
}
run_pmc () {     # name, counters, command...
  name=$1; ctrs=$2; shift; shift
  rm -rf /tmp/pmc_$name
  rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_$name -o $name -- "$@" > $out/${tag}_${name}_pmc.log 2>&1
  python $root/tools/pmc_summary.py $(find /tmp/pmc_$name -name "*.db" | head -1) > $out/${tag}_${name}.txt 2>&1
}
BENCH="python $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-model-step"
run_stats bench $BENCH
grep '"metric"' $out/${tag}_bench.log > $out/${tag}_bench_under_rocprof.json
run_pmc pmc_FETCH_SIZE "FETCH_SIZE" $BENCH
run_pmc pmc_WRITE_SIZE "WRITE_SIZE" $BENCH
run_pmc pmc_SQ_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" $BENCH
run_pmc pmc_SQ_lds "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" $BENCH
ITER="python $root/tools/dev/dev_iter_profile.py 20"
run_stats iter $ITER
run_pmc iter_pmc_FETCH_SIZE "FETCH_SIZE" $ITER
run_pmc iter_pmc_WRITE_SIZE "WRITE_SIZE" $ITER
NATIVE="python $root/tools/dev/native_overhead.py 300000 40"
run_stats native300k $NATIVE
grep "native trainer" $out/${tag}_native300k.log >> $out/${tag}_native300k_kernel_stats.txt
KNN="python $root/tools/dev/dev_knn_time.py"
run_stats knn $KNN
run_pmc knn_pmc_FETCH_SIZE "FETCH_SIZE" $KNN
run_pmc knn_pmc_WRITE_SIZE "WRITE_SIZE" $KNN
cd $root
python tools/pmc_traffic.py $out/${tag}_pmc_FETCH_SIZE.txt $out/${tag}_pmc_WRITE_SIZE.txt $out/${tag}_pmc_SQ_valu.txt $out/${tag}_bench_kernel_stats.txt $out/${tag}_pmc_traffic.json \
    $out/${tag}_iter_pmc_FETCH_SIZE.txt $out/${tag}_iter_pmc_WRITE_SIZE.txt $out/${tag}_iter_kernel_stats.txt $out/${tag}_knn_pmc_FETCH_SIZE.txt $out/${tag}_knn_pmc_WRITE_SIZE.txt $out/${tag}_knn_kernel_stats.txt | tail -60
# the default bench line quotes these counters (bench.py: PMC_FILE, guarded by the kernel sources' hashes): put them where it looks
cp $out/${tag}_pmc_traffic.json $root/profiles/${tag}_pmc_traffic.json
python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err
head -16 $out/${tag}_bench_kernel_stats.txt | cut -c1-70,161-250
