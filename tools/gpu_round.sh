#!/bin/bash
# One GPU session: parity tests, smoke, bench, rocprofv3 kernel stats (+ optional PMC passes).
# Usage: tools/gpu_round.sh [tests] [bench] [prof] [pmc]
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for what in "$@"; do
case $what in
tests)
  timeout 1500 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
  tail -15 $OUT/pytest_gpu.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt
  tail -3 $OUT/smoke.txt ;;
bench)
  timeout 900 python bench.py > $OUT/bench.txt 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
  cat $OUT/bench.txt; tail -5 $OUT/bench.err ;;
prof)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_bench.log 2>&1)
  find $OUT/prof_bench -name '*kernel_stats.csv' | head -1 | xargs -r head -40 ;;
pmc)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o ntt -- python $REPO/bench.py --ntt-only > $OUT/pmc_fetch.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o ntt -- python $REPO/bench.py --ntt-only > $OUT/pmc_write.log 2>&1)
  ls -R $OUT/pmc_fetch | head -20 ;;
esac
done
