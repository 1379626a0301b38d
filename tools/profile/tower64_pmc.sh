#!/bin/bash
# Counters under the 64-channel tower (BASELINE configs 4/5): gpurun -- 'bash tools/profile/tower64_pmc.sh'
export TMPDIR=/tmp; O=gpurun_out/tower64; mkdir -p $O
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -- python3 tools/tower_probe.py 64 8 2048 > $O/p$i.log 2>&1
  python tools/profile/summarize_pmc_all.py $O/p$i 8 c4_conv_tower > $O/p$i.json 2>> $O/p$i.log
  rm -rf $O/p$i
  i=$((i+1))
done
cat $O/p*.json
