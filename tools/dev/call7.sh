set -x
bash tools/pmc.sh c7a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" > /dev/null
bash tools/pmc.sh c7b "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" > /dev/null
bash tools/pmc.sh c7F "FETCH_SIZE" > /dev/null
bash tools/pmc.sh c7W "WRITE_SIZE" > /dev/null
bash tools/pmc.sh c7T "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" > /dev/null
for t in c7a c7b c7F c7W c7T; do echo "== $t"; head -12 gpurun_out/${t}_pmc.txt | cut -c1-260; done
