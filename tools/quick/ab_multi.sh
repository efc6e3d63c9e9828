#!/bin/bash
# headline bench over several library variants (seal_amd/lib/variants/NAME.so), ROUNDS rounds, alternating
set -u
ROUNDS=${ROUNDS:-2}
cp seal_amd/lib/libsealhip.so /tmp/keep.so
for r in $(seq $ROUNDS); do for v in "$@"; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); c=j['roofline_configs1']['chains']
print('%-8s %8.1f ct/s %7.3f ms | NTT %7.1f | C2 fwd %7.1f inv %7.1f' % ('$v$r', j['value'], j['ms_per_step'], j['roofline']['achieved'], c[0]['forward']['achieved'], c[0]['inverse']['achieved']))"
done; done
cp /tmp/keep.so seal_amd/lib/libsealhip.so
