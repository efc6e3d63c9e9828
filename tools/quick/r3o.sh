#!/bin/bash
# round 3: BEHZ with limbs split on the host (behz4) - BFV tests, then A/B on BFV configs[3]
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r3q; mkdir -p $O
cp seal_amd/lib/variants/behz4.so seal_amd/lib/libsealhip.so
(timeout 900 python -m pytest tests -m gpu -x -q -k "bfv or rns or config4 or golden or fuzz or dropin" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt); tail -3 $O/pytest.txt
for r in 1 2; do for v in cur behz4; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  c4=$(timeout 300 python bench.py --workload bfv_c4 --steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
  echo "$v$r C4 ct/s, ms/step: $c4"
done; done 2>&1 | tee $O/ab_c4.txt
cp seal_amd/lib/variants/behz4.so seal_amd/lib/libsealhip.so
