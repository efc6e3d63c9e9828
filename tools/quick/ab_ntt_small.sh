#!/bin/bash
# A/B of the single-launch transforms at N = 2^13 / 2^14: default (second generation), first generation (forward N = 2^13
# only, two-launch inverse), two-launch engine everywhere
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
for spec in "v2:" "v1:SEALHIP_NTT_FUSED_V1=1" "twopass:SEALHIP_NTT_NOFUSED=1"; do
  label=${spec%%:*}; envs=${spec#*:}
  echo "== $label ($envs)"
  env $envs timeout 300 python tools/quick/ntt_perf2.py 2>&1 | tee $OUT/ntt_small_$label.txt
done
