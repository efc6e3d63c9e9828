#!/bin/bash
# round 3: double-precision inverse single-launch kernel with / without prefetch and twiddle variants
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r3k; mkdir -p $O
ROUNDS=3 tools/quick/ab_multi.sh cur curab fpnopf 2>&1 | tee $O/ab_multi.txt
