#!/bin/bash
# same-box A/B of the headline step: seal_amd/lib/variants/{pre,now}.so alternating, then the 2^16 key-switch parity tests with `now`
set -u
export TMPDIR=/tmp
cp seal_amd/lib/libsealhip.so /tmp/keep.so
for r in 1 2 3; do for v in pre now; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  hl=$(timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
  echo "$v$r headline ct/s, ms/step: $hl"
done; done
cp /tmp/keep.so seal_amd/lib/libsealhip.so
mkdir -p gpurun_out/r3w
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_serialization.py -x -q -k "65536 or headline or key_save or lean or sampled" 2>&1 | tail -4 | tee gpurun_out/r3w/pytest.txt
