#!/bin/bash
# randomised operation sequences against the reference on the device: many more seeds than the test suite runs
set -u
export TMPDIR=/tmp
O=gpurun_out/r3t; mkdir -p $O
FUZZ_SEED0=300 FUZZ_SEEDS=10 timeout 900 python tools/quick/fuzz_stress.py 2>&1 | tail -5 | tee $O/fuzz_plain.txt
FUZZ_SEED0=400 FUZZ_SEEDS=6 FUZZ_WILD=0.25 timeout 600 python tools/quick/fuzz_stress.py 2>&1 | tail -5 | tee $O/fuzz_wild.txt
