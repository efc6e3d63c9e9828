#!/bin/bash
# round 3, sixth device session: Shoup keys + 64-bit sums in the integer ks2 (with / without prefetch), sequential single-launch runs
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r3f; mkdir -p $O
cp seal_amd/lib/variants/ksA.so seal_amd/lib/libsealhip.so
(timeout 1200 python -m pytest tests -m gpu -x -q -rs > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt); tail -4 $O/pytest.txt
ROUNDS=2 tools/quick/ab_multi.sh base ksA ksB 2>&1 | tee $O/ab_multi.txt
for r in 1 2; do for v in base ksA ksB; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  c4=$(timeout 300 python bench.py --workload bfv_c4 --steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'])")
  echo "$v$r C4 ct/s: $c4"
done; done 2>&1 | tee $O/ab_c4.txt
cp seal_amd/lib/variants/ksA.so seal_amd/lib/libsealhip.so
for r in 1 2; do timeout 300 python tools/bench_configs.py --configs C2 --no-cpu 2>/dev/null | grep fwd_GBs | cut -c1-300; done | tee $O/c2.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/prof -o hl -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-verify > $O/prof.log 2>&1)
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $O/headline_kernel_stats.txt 2>&1; head -24 $O/headline_kernel_stats.txt
python tools/step_timeline.py $DB > $O/headline_step_timeline.txt 2>&1; tail -30 $O/headline_step_timeline.txt
rm -rf $O/prof
