#!/bin/bash
# Regenerates the measurements behind profiles/r02_* on one MI355X (run from the repo root on the GPU box, e.g. through
# `gpurun -- tools/repro_profiles.sh all`).  Output goes to gpurun_out/repro/; copy what you want judged into profiles/.
#   tests     pytest -m gpu + smoke                       -> pytest.txt, smoke.txt
#   bench     python bench.py (CPU baseline + live PMC)    -> bench.json
#   trace     rocprofv3 --kernel-trace of a short bench    -> kernel_stats.txt, step_timeline.txt
#   counters  PMC tables of the NTT launch and the ks kernels (one rocprofv3 --pmc pass per counter group; no trace domains)
#   configs   the other BASELINE configurations            -> configs.txt
set -u
export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out/repro; mkdir -p $O
[ $# -eq 0 ] && set -- all
for what in "$@"; do
  case $what in
  tests|all)
    (timeout 1500 python -m pytest tests -m gpu -q -rs > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt); tail -3 $O/pytest.txt
    (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt); tail -1 $O/smoke.txt ;;&
  bench|all)
    timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-220 $O/bench.json ;;&
  trace|all)
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/prof -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $O/prof.log 2>&1)
    DB=$(find $O/prof -name "*.db" | head -1)
    python tools/rocpd_summary.py $DB > $O/kernel_stats.txt; python tools/step_timeline.py $DB > $O/step_timeline.txt; rm -rf $O/prof
    tail -3 $O/step_timeline.txt ;;&
  counters|all)
    timeout 600 python tools/pmc_table.py > $O/ntt_counters.txt 2> $O/ntt_counters.err
    timeout 600 python tools/pmc_table.py --bench-args "--batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-verify" --filter ks > $O/ks_counters.txt 2> $O/ks_counters.err
    wc -l $O/ntt_counters.txt $O/ks_counters.txt ;;&
  configs|all)
    (timeout 300 python tools/bench_configs.py --configs C2,C3 --no-cpu; for w in bfv_c4 rotate_c5; do timeout 400 python bench.py --workload $w --steps 6 --warmup 2 --no-pmc | tail -1; done) > $O/configs.txt 2> $O/configs.err
    cut -c1-200 $O/configs.txt ;;
  esac
done
