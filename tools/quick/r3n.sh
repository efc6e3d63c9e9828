#!/bin/bash
# round 3: last-stage twiddles of the integer ks2 in LDS (SEALHIP_KS2_INT_TWB3)
set -u
export TMPDIR=/tmp
for r in 1 2; do for v in notwb3 twb3; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  c4=$(timeout 300 python bench.py --workload bfv_c4 --steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
  hl=$(timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'])")
  echo "$v$r C4 ct/s, ms/step: $c4 | headline $hl"
done; done
cp seal_amd/lib/variants/twb3.so seal_amd/lib/libsealhip.so
