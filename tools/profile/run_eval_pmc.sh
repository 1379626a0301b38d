#!/bin/bash
# Counters UNDER THE EVALUATOR (since round 3): run on the GPU box from the repo root; merge with tools/profile/merge_eval_pmc.py.
#   bash tools/profile/run_eval_pmc.sh [M=2048] [tag]
# rocprofv3 --pmc passes over tools/evaluator_probe.py (program directly after "--"), one pass per
# counter group, for the hand GEMM and for hipBLASLt.
M=${1:-2048}; TAG=${2:-r06}; O=gpurun_out/${TAG}_evalpmc_M$M; mkdir -p $O; export TMPDIR=/tmp
rocprofv3 -L > $O/counters_available.txt 2>&1
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum"
 "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_VMEM"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum"
 "TCC_BUSY_avr TCC_REQ_sum"
)
for BK in "hip 0" "hipblaslt 0" ${EXTRA_BACKENDS}; do
  set -- $BK; B=$1; C=$2
  i=0
  for P in "${PASSES[@]}"; do
    rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/${B}${C}_p$i -- python3 tools/evaluator_probe.py $M $B $C > $O/${B}${C}_p$i.log 2>&1
    python tools/profile/summarize_pmc_all.py $O/${B}${C}_p$i 8 > $O/${B}${C}_p$i.json 2>> $O/${B}${C}_p$i.log
    rm -rf $O/${B}${C}_p$i
    i=$((i+1))
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${B}${C}_stats -- python3 tools/evaluator_probe.py $M $B $C 4 32 30 > $O/${B}${C}_stats.log 2>&1
  cp $(find $O/${B}${C}_stats -name '*kernel_stats.csv' | head -1) $O/${B}${C}_kernel_stats.csv 2>/dev/null
  rm -rf $O/${B}${C}_stats
done
ls -la $O
