#!/bin/bash
# round 3, third device session: single-launch integer transforms at 2^13 / 2^14
set -u
export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q -rs > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt); tail -4 $O/pytest.txt
for r in 1 2; do
  for e in "X=1" "SEALHIP_NTT_NOFUSED_INT=1"; do
    echo "== $e"; env $e timeout 300 python tools/bench_configs.py --configs C2 --no-cpu 2>/dev/null | grep fwd_GBs | cut -c1-400
  done
done 2>&1 | tee $O/c2.txt
ROUNDS=1 tools/quick/ab_multi.sh base fint 2>&1 | tee $O/ab_multi.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/prof -o c2 -- python /root/repo/tools/bench_configs.py --configs C2 --no-cpu > $O/prof.log 2>&1)
DB=$(find /root/repo/$O/prof $O/prof -name "*.db" 2>/dev/null | head -1)
python tools/rocpd_summary.py $DB > $O/c2_kernel_stats.txt 2>&1; head -30 $O/c2_kernel_stats.txt
python tools/quick/dump_dispatches.py $DB fused2 40 > $O/c2_dispatches.txt 2>&1; tail -20 $O/c2_dispatches.txt
rm -rf $O/prof
