#!/bin/bash
# round 3: mixed chain at N = 2^14 (C3) and the per-size NTT table, round-2 base against the current build
set -u
export TMPDIR=/tmp
cp seal_amd/lib/libsealhip.so /tmp/keep.so
for r in 1 2; do for v in base cur; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  echo "$v$r $(timeout 300 python tools/bench_configs.py --configs C3 --no-cpu 2>/dev/null | grep -E '^\{' | cut -c1-260)"
  timeout 300 python tools/quick/ntt_sizes.py 2>/dev/null | sed "s/^/$v$r /"
done; done
cp /tmp/keep.so seal_amd/lib/libsealhip.so
