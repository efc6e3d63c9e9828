#!/bin/bash
# round 3: same-box A/B of seal_amd/lib/variants/{pre,now}.so on BFV configs[3] and the headline
set -u
export TMPDIR=/tmp
cp seal_amd/lib/libsealhip.so /tmp/keep.so
for r in 1 2; do for v in pre now; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  c4=$(timeout 300 python bench.py --workload bfv_c4 --steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
  hl=$(timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'])")
  echo "$v$r C4 ct/s, ms/step: $c4 | headline $hl"
done; done
cp /tmp/keep.so seal_amd/lib/libsealhip.so
