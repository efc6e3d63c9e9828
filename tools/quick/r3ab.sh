#!/bin/bash
# same-box A/B of the headline and the rotate workload: seal_amd/lib/variants/{pre,now}.so; then the tail parity tests with `now`
set -u
export TMPDIR=/tmp
O=gpurun_out/r3ab; mkdir -p $O
cp seal_amd/lib/libsealhip.so /tmp/keep.so
for r in 1 2 3; do for v in pre now; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  hl=$(timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
  ro=$(timeout 300 python bench.py --workload rotate_c5 --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
  echo "$v$r headline ct/s, ms/step: $hl | rotate_c5 $ro"
done; done 2>&1 | tee $O/ab.txt
cp /tmp/keep.so seal_amd/lib/libsealhip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "65536 or headline or lean or sampled or tail or rescale or multi_level or pipeline" > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -2
