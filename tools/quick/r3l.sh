#!/bin/bash
# round 3: double-precision forward single-launch kernel without prefetch at 2^13
set -u
export TMPDIR=/tmp
cp seal_amd/lib/libsealhip.so /tmp/keep.so
for r in 1 2 3; do for v in cur fwdnopf; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); c=j['roofline_configs1']['chains']
print('$v$r  mixed fwd %7.1f inv %7.1f | all-fp fwd %7.1f inv %7.1f' % (c[0]['forward']['achieved'], c[0]['inverse']['achieved'], c[1]['forward']['achieved'], c[1]['inverse']['achieved']))"
done; done
cp /tmp/keep.so seal_amd/lib/libsealhip.so
