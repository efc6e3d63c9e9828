#!/bin/bash
# round 3, fourth device session: variants of the single-launch integer transforms, carry-free BEHZ dot products
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r3d; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q -k "bfv or rns or config4 or single_launch or golden" > $O/pytest_bfv.txt 2>&1; echo "rc=$?" >> $O/pytest_bfv.txt); tail -3 $O/pytest_bfv.txt
ROUNDS=2 tools/quick/ab_multi.sh fintA fintB fintC 2>&1 | tee $O/ab_multi.txt
for r in 1 2; do for v in base fintA; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  c4=$(timeout 300 python bench.py --workload bfv_c4 --steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'])")
  echo "$v$r C4 ct/s: $c4"
done; done 2>&1 | tee $O/ab_c4.txt
cp seal_amd/lib/variants/fintA.so seal_amd/lib/libsealhip.so
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/prof -o c2 -- python $R/tools/bench_configs.py --configs C2 --no-cpu > $O/prof.log 2>&1)
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $O/c2_kernel_stats.txt 2>&1; head -12 $O/c2_kernel_stats.txt
python tools/quick/dump_dispatches.py $DB ntt2 400 2>&1 | tail -24 > $O/c2_dispatches.txt; cat $O/c2_dispatches.txt
rm -rf $O/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/prof4 -o c4 -- python $R/bench.py --workload bfv_c4 --total-batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-verify > $O/prof4.log 2>&1)
DB=$(find $O/prof4 -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $O/c4_kernel_stats.txt 2>&1; head -26 $O/c4_kernel_stats.txt
rm -rf $O/prof4
