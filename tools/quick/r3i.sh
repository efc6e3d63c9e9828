#!/bin/bash
# round 3: workgroup -> (tile, transform) mapping that gives the eight XCDs different transforms
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r3i; mkdir -p $O
ROUNDS=2 tools/quick/ab_multi.sh nospread spread 2>&1 | tee $O/ab_multi.txt
for r in 1 2; do for v in nospread spread; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  c4=$(timeout 300 python bench.py --workload bfv_c4 --steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['achieved'])")
  echo "$v$r C4 ct/s, NTT GB/s: $c4"
done; done 2>&1 | tee $O/ab_c4.txt
cp seal_amd/lib/variants/spread.so seal_amd/lib/libsealhip.so
(timeout 900 python -m pytest tests -m gpu -x -q -k "ntt or pipeline or bfv or golden" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt); tail -3 $O/pytest.txt
