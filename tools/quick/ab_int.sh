#!/bin/bash
# A/B of library variants on the integer-back-end workloads: configs[1] NTT (C2) and BFV configs[3] (C4)
# usage: tools/quick/ab_int.sh VARIANT [ROUNDS]   (seal_amd/lib/variants/VARIANT.so against variants/base.so)
set -u
V=$1; ROUNDS=${2:-2}
cp seal_amd/lib/libsealhip.so /tmp/keep.so
for r in $(seq $ROUNDS); do for v in base $V; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  c2=$(timeout 300 python tools/bench_configs.py --configs C2 --no-cpu 2>/dev/null | grep fwd_GBs | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['fwd_GBs'], j['inv_GBs'])")
  c4=$(timeout 300 python bench.py --workload bfv_c4 --steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'])")
  echo "$v$r  C2 fwd/inv GB/s: $c2   C4 ct/s: $c4"
done; done
cp /tmp/keep.so seal_amd/lib/libsealhip.so
