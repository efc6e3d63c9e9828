#!/bin/bash
# Round-6 measurements on one MI355X (run from the repo root on the GPU box: `gpurun -- tools/gpu_round6.sh <sections>`).
# Output goes to gpurun_out/r06/; what is to be judged is copied into profiles/r06_* afterwards.
#   mall      tools/microbench/mall_bw: what the Infinity Cache gives a re-used footprint (profiles/r06_mall_bw.txt)
#   handoff   tools/microbench/ring_handoff: the hand-over protocols of a one-launch two-pass transform (r06_ring_handoff.txt)
#   ring      the real one-launch kernels (SEALHIP_NTT_RING=1 / 2) against the two launches: parity test + leg (r06_ring_kernel.txt)
#   latency   the headline step at batch 1 / 2 / 4 / 8, eager and as a graph replay, with the batch-1 time line (r06_latency.txt)
#   lazy      same-box A/B of the deferred tensor product (SEALHIP_LAZY_PRODUCT=0 against the default) + the step's time line (r06_lazy_product.txt)
#   prev      same-box A/B against a variant library built from another commit's kernels (seal_amd/lib/variants/prev.so; see the
#             recipe in profiles/r06_lazy_product.txt, table 4)
#   fuzz      tools/quick/fuzz_deferred.py, FUZZ_SECONDS (default 300): random sequences with deferred tails and pending products
#   multi     the two- and eight-process tests that share the one GPU (r06_pytest_multi.txt)
#   tests     pytest -m gpu + smoke
#   bench     python bench.py (default line)
#   trace     rocprofv3 --kernel-trace --stats of a short bench + the step's time line
# Microbenchmarks are built on the box: hipcc -O3 --offload-arch=gfx950 tools/microbench/X.hip -o tools/microbench/X
set -u
export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out/r06; mkdir -p $O
export SEALHIP_ABORT_TRACE=$O/abort_trace.txt
common="--no-cpu-baseline --no-pmc --no-verify --no-children"
mb() { [ -x tools/microbench/$1 ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/microbench/$1.hip -o tools/microbench/$1; }
[ $# -eq 0 ] && set -- tests bench trace
for what in "$@"; do
  echo "=== $what $(date +%T)"
  case $what in
  mall)
    mb mall_bw; timeout 600 tools/microbench/mall_bw ab > $O/mall_bw.txt 2>&1; tail -60 $O/mall_bw.txt ;;
  handoff)
    mb ring_handoff; timeout 600 tools/microbench/ring_handoff > $O/ring_handoff.txt 2>&1; tail -60 $O/ring_handoff.txt ;;
  ring)
    timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "ntt_ring" > $O/pytest_ring.txt 2>&1; tail -2 $O/pytest_ring.txt
    tools/ab.sh --rounds 2 --workload ntt --out gpurun_out/r06/ab_ring two_launches:default ring1:default:SEALHIP_NTT_RING=1 ring2:default:SEALHIP_NTT_RING=2 ;;
  latency)
    for b in 1 2 4 8; do
      for g in "" "--graph"; do
        timeout 300 python bench.py --batch $b --steps 400 --warmup 20 $g $common 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('batch $b %-8s %.4f ms/step  %.1f ct/s' % ('$g' or 'eager', j['ms_per_step'], j['value']))"
      done
    done 2>&1 | tee $O/latency.txt
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/prof_b1 -o t -- python $REPO/bench.py --batch 1 --steps 5 --warmup 2 $common > $O/prof_b1.log 2>&1)
    python tools/step_timeline.py $(find $O/prof_b1 -name "*.db" | head -1) > $O/timeline_batch1.txt 2>&1; rm -rf $O/prof_b1; tail -30 $O/timeline_batch1.txt ;;
  lazy)
    tools/ab.sh --rounds ${ROUNDS:-3} --trace --out gpurun_out/r06/ab_lazy base:default:SEALHIP_LAZY_PRODUCT=0 lazy:default ;;
  prev)
    tools/ab.sh --rounds ${ROUNDS:-4} --out gpurun_out/r06/ab_prev prev:prev new:default ;;
  fuzz)
    FUZZ_SECONDS=${FUZZ_SECONDS:-300} FUZZ_SEED0=${FUZZ_SEED0:-900} timeout 1200 python tools/quick/fuzz_deferred.py > $O/fuzz_deferred.txt 2>&1; tail -4 $O/fuzz_deferred.txt ;;
  multi)
    timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -q --durations=8 > $O/pytest_multi.txt 2>&1; tail -4 $O/pytest_multi.txt ;;
  tests)
    (timeout 1800 python -m pytest tests -m gpu -q -rs --durations=8 > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt); tail -6 $O/pytest.txt
    (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt); tail -1 $O/smoke.txt ;;
  bench)
    timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json ;;
  trace)
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $REPO/bench.py --steps 3 --warmup 1 $common > $O/prof.log 2>&1)
    DB=$(find $O/prof -name "*.db" | head -1)
    python tools/rocpd_summary.py $DB > $O/rocprof_bench_kernel_stats.txt; python tools/step_timeline.py $DB > $O/step_timeline.txt
    find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/rocprofv3_stats_kernel_stats.csv; rm -rf $O/prof
    tail -40 $O/step_timeline.txt ;;
  esac
done
[ -s $O/abort_trace.txt ] && { echo "ABORT TRACE:"; cat $O/abort_trace.txt; }
exit 0
