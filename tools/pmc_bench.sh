#!/bin/bash
# rocprofv3 PMC passes over one short bench.py run (counter groups in separate passes; kernel-trace only)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_bench; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o r -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/g$i.log 2>&1)
  tail -2 $OUT/g$i.log | head -1
done
