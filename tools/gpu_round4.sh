#!/bin/bash
# Round-4 measurements on one MI355X (run from the repo root on the GPU box: `gpurun -- tools/gpu_round4.sh <sections>`).
# Output goes to gpurun_out/r04/; what is to be judged is copied into profiles/ afterwards.
#   order     tools/microbench/d2d_order: does the runtime order copy -> memset -> in-place kernel -> download on one stream?
#   soak      tests/soak_cases.py for SOAK_SECONDS (default 150) in every host-copy mode
#   tests     pytest -m gpu + smoke
#   bench     python bench.py (default line: headline + the configs[3] / configs[4] children)
#   trace     rocprofv3 --kernel-trace --stats of a short bench + the step's time line
set -u
export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out/r04; mkdir -p $O
export SEALHIP_ABORT_TRACE=$O/abort_trace.txt
[ $# -eq 0 ] && set -- order soak tests bench
for what in "$@"; do
  case $what in
  order)
    for mode in 0 1 2; do for stream in 0 1; do
      timeout 120 tools/microbench/d2d_order ${ORDER_ITERS:-20000} $mode $stream 2>&1 | tail -4
    done; done > $O/d2d_order.txt; cat $O/d2d_order.txt ;;
  soak)
    timeout 900 python - > $O/soak.txt 2>&1 <<PY
import os, sys, json
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import seal_amd as S
S.load()
import soak_cases as K
try:
    st = K.run_soak(float(os.environ.get("SOAK_SECONDS", "150")), seed=int(os.environ.get("SOAK_SEED", "1")), dump_dir="$O")
    print("soak clean:", json.dumps(st))
except AssertionError as e:
    print("SOAK MISMATCH:", e)
PY
    tail -3 $O/soak.txt ;;
  tests)
    (timeout 1500 python -m pytest tests -m gpu -q -rs > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt); tail -4 $O/pytest.txt
    (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt); tail -1 $O/smoke.txt ;;
  bench)
    timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json ;;
  trace)
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-verify --no-children > $O/prof.log 2>&1)
    DB=$(find $O/prof -name "*.db" | head -1)
    python tools/rocpd_summary.py $DB > $O/rocprof_bench_kernel_stats.txt; python tools/step_timeline.py $DB > $O/step_timeline.txt
    find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/rocprofv3_stats_kernel_stats.csv; rm -rf $O/prof
    tail -24 $O/step_timeline.txt ;;
  esac
done
[ -s $O/abort_trace.txt ] && { echo "ABORT TRACE:"; cat $O/abort_trace.txt; }
exit 0
