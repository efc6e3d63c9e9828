#!/bin/bash
# round 3: the two class runs of the single-launch kernels at N = 2^14 side by side or in sequence (SEALHIP_FUSED_FORK14)
set -u
export TMPDIR=/tmp
cp seal_amd/lib/libsealhip.so /tmp/keep.so
for r in 1 2; do for v in nofork14 fork14; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  timeout 300 python tools/quick/ntt_sizes.py 2>/dev/null | grep 16384 | sed "s/^/$v$r /"
  echo "$v$r $(timeout 300 python tools/bench_configs.py --configs C3 --no-cpu 2>/dev/null | grep -E '^\{' | cut -c60-200)"
done; done
cp /tmp/keep.so seal_amd/lib/libsealhip.so
