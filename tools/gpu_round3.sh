#!/bin/bash
# Regenerates the measurements behind profiles/r03_* on one MI355X (run from the repo root on the GPU box:
# `gpurun -- tools/gpu_round3.sh all`).  Output goes to gpurun_out/r03/; copy what you want judged into profiles/.
#   tests     pytest -m gpu + smoke                                   -> pytest.txt, smoke.txt
#   bench     python bench.py for the three workloads (CPU baseline + live PMC)  -> bench_*.json
#   trace     rocprofv3 --kernel-trace --stats of a short bench        -> rocprof_bench_kernel_stats.txt, step_timeline.txt
#   counters  PMC tables: the 2^16 NTT launch, the key-switch kernels, the single-launch integer transforms
#   configs1  dispatch time line of the configs[1] chain {60,40,40,60}
#   copy      tools/microbench/copy_footprint
set -u
export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out/r03; mkdir -p $O
[ $# -eq 0 ] && set -- all
for what in "$@"; do
  case $what in
  tests|all)
    (timeout 1500 python -m pytest tests -m gpu -q -rs > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt); tail -3 $O/pytest.txt
    (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt); tail -1 $O/smoke.txt ;;&
  bench|all)
    timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
    for w in bfv_c4 rotate_c5; do timeout 900 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"; cut -c1-200 $O/bench_$w.json; done ;;&
  trace|all)
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-verify > $O/prof.log 2>&1)
    DB=$(find $O/prof -name "*.db" | head -1)
    python tools/rocpd_summary.py $DB > $O/rocprof_bench_kernel_stats.txt; python tools/step_timeline.py $DB > $O/step_timeline.txt
    find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/rocprofv3_stats_kernel_stats.csv; rm -rf $O/prof
    tail -22 $O/step_timeline.txt ;;&
  counters|all)
    timeout 600 python tools/pmc_table.py --groups 0,1,7 > $O/ntt_counters.txt 2> $O/ntt_counters.err
    timeout 600 python tools/pmc_table.py --bench-args "--batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-verify" --filter ks --groups 7,8 > $O/ks_counters.txt 2> $O/ks_counters.err
    timeout 600 python tools/pmc_table.py --bench-args "--batch 8 --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-verify" --filter "fused2<5" --groups 0,1,7,8 > $O/configs1_counters.txt 2> $O/configs1_counters.err
    wc -l $O/ntt_counters.txt $O/ks_counters.txt $O/configs1_counters.txt ;;&
  configs1|all)
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/prof1 -o c2 -- python $REPO/tools/bench_configs.py --configs C2 --no-cpu > $O/configs1.log 2>&1)
    DB=$(find $O/prof1 -name "*.db" | head -1)
    python tools/quick/dump_dispatches.py $DB fused2 400 2>&1 | tail -16 > $O/configs1_dispatches.txt; grep fwd_GBs $O/configs1.log >> $O/configs1_dispatches.txt; rm -rf $O/prof1
    cat $O/configs1_dispatches.txt ;;&
  copy|all)
    tools/microbench/copy_footprint > $O/microbench_copy_footprint.txt 2>&1; cat $O/microbench_copy_footprint.txt ;;
  esac
done
