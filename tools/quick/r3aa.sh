#!/bin/bash
# same-box A/B: seal_amd/lib/variants/{pre,now}.so on BFV configs[3] and the headline, then the key-switch parity tests with `now`
set -u
export TMPDIR=/tmp
O=gpurun_out/r3aa; mkdir -p $O
cp seal_amd/lib/libsealhip.so /tmp/keep.so
for r in 1 2 3 4; do for v in pre now; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  c4=$(timeout 300 python bench.py --workload bfv_c4 --steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
  hl=$(timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
  echo "$v$r C4 ct/s, ms/step: $c4 | headline $hl"
done; done 2>&1 | tee $O/ab.txt
cp /tmp/keep.so seal_amd/lib/libsealhip.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_serialization.py -x -q -k "65536 or headline or lean or sampled or bfv or rotate or digit or relin or multi_level" > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -2
