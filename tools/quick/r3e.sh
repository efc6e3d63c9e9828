#!/bin/bash
# round 3, fifth device session: single-launch integer transforms - twiddle / prefetch variants, fork vs no fork; BFV without copies
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r3e; mkdir -p $O
c2() { timeout 300 python tools/bench_configs.py --configs C2 --no-cpu 2>/dev/null | grep fwd_GBs | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['fwd_GBs'], j['inv_GBs'])"; }
for r in 1 2; do for v in v1 v2 v3; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  echo "$v$r fork   C2(8192 polys) fwd/inv: $(c2)"
  echo "$v$r nofork C2(8192 polys) fwd/inv: $(SEALHIP_NTT_NOFORK=1 c2)"
done; done 2>&1 | tee $O/c2_variants.txt
ROUNDS=1 tools/quick/ab_multi.sh v1 v2 v3 2>&1 | tee $O/ab_multi.txt
cp seal_amd/lib/variants/v1.so seal_amd/lib/libsealhip.so
(timeout 900 python -m pytest tests -m gpu -x -q -k "bfv or rns or config4 or single_launch or golden or fuzz" > $O/pytest_bfv.txt 2>&1; echo "rc=$?" >> $O/pytest_bfv.txt); tail -3 $O/pytest_bfv.txt
for r in 1 2; do for v in base v1; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  c4=$(timeout 300 python bench.py --workload bfv_c4 --steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'])")
  echo "$v$r C4 ct/s: $c4"
done; done 2>&1 | tee $O/ab_c4.txt
cp seal_amd/lib/variants/v1.so seal_amd/lib/libsealhip.so
