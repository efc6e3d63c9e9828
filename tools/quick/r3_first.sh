#!/bin/bash
# round 3, first device session: parity tests of the new integer butterflies, then base (round-2 final) against new
set -u
export TMPDIR=/tmp
O=gpurun_out/r3a; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q -rs > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt); tail -4 $O/pytest.txt
ROUNDS=2 tools/quick/ab_multi.sh base new 2>&1 | tee $O/ab_multi.txt
tools/quick/ab_int.sh new 2 2>&1 | tee $O/ab_int.txt
