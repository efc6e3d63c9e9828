#!/bin/bash
# configs1 with library variants: /tmp/c1var.sh ROUNDS NAME...
export TMPDIR=/tmp
cp seal_amd/lib/libsealhip.so /tmp/keep.so
R=$1; shift
for r in $(seq 1 $R); do for v in "$@"; do
  if [ $v = default ]; then cp /tmp/keep.so seal_amd/lib/libsealhip.so; else cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so; fi
  tools/quick/configs1_ab.sh 1 $v | sed "s/round 1/round $r/"
done; done
cp /tmp/keep.so seal_amd/lib/libsealhip.so
