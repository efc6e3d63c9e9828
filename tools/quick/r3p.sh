#!/bin/bash
# round 3: kernel table of BFV configs[3] at batch 256 (profiles/r03_bfv_c4_kernel_stats.txt)
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r3p; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/prof4 -o c4 -- python $R/bench.py --workload bfv_c4 --total-batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-verify > $O/prof4.log 2>&1)
DB=$(find $O/prof4 -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $O/c4_kernel_stats.txt 2>&1; head -12 $O/c4_kernel_stats.txt
rm -rf $O/prof4
timeout 600 python tools/pmc_table.py --bench-args "--workload bfv_c4 --total-batch 64 --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-verify" --filter behz --groups 7,8 > $O/behz_counters.txt 2> $O/behz_counters.err; grep -E "avg duration|SQ_INSTS_VALU |SQ_INSTS_SALU|SQ_WAIT_INST_ANY  |SQ_ACTIVE_INST_VALU  |SQ_INSTS_SMEM" $O/behz_counters.txt
