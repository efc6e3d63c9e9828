#!/bin/bash
# rocprofv3 PMC passes over the NTT kernels (one counter group per pass; no tracing domains besides kernel-trace)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_ntt; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o r -- python $REPO/tools/quick/ntt_one.py "$@" > $OUT/g$i.log 2>&1)
  tail -2 $OUT/g$i.log | head -1
done
