#!/bin/bash
# Shoup products as two asm blocks (SEALHIP_INT_BLOCKS=1, variants/blk1.so) against one instruction per statement (blk0.so): device field
# check + integer-heavy parity tests with blk1, then same-box A/B on BFV configs[3], the headline and the configs[1] chain
set -u
export TMPDIR=/tmp
O=gpurun_out/r3z; mkdir -p $O
cp seal_amd/lib/libsealhip.so /tmp/keep.so
cp seal_amd/lib/variants/${NEW:-blk1}.so seal_amd/lib/libsealhip.so
timeout 120 seal_amd/lib/device_field_check > $O/device_field_check.txt 2>&1; echo "device_field_check rc=$?"; tail -2 $O/device_field_check.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bfv or bgv or fused or field or 65536 or multi_level" > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -2
for r in 1 2; do for v in blk0 ${NEW:-blk1}; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  c4=$(timeout 300 python bench.py --workload bfv_c4 --steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
  hl=$(timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'])")
  c2=$(timeout 300 python tools/bench_configs.py --configs C2 --no-cpu 2>/dev/null | grep fwd_GBs | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['fwd_GBs'], j['inv_GBs'])")
  echo "$v$r C4 ct/s, ms/step: $c4 | headline $hl | configs[1] chain fwd/inv GB/s: $c2"
done; done 2>&1 | tee $O/ab.txt
cp /tmp/keep.so seal_amd/lib/libsealhip.so
