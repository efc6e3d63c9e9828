#!/bin/bash
# round 3, second device session: the instruction wrappers (asm1) - device check, parity, A/B against base and the first cut
set -u
export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
seal_amd/lib/device_field_check 2>&1 | tee $O/device_field_check.txt
(timeout 1200 python -m pytest tests -m gpu -x -q -rs > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt); tail -4 $O/pytest.txt
ROUNDS=2 tools/quick/ab_multi.sh base asm1 2>&1 | tee $O/ab_multi.txt
for r in 1 2; do for v in base asm1; do
  cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
  c4=$(timeout 300 python bench.py --workload bfv_c4 --steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'])")
  echo "$v$r C4 ct/s: $c4"
done; done 2>&1 | tee $O/ab_c4.txt
cp seal_amd/lib/variants/asm1.so seal_amd/lib/libsealhip.so
timeout 600 python tools/pmc_table.py --bench-args "--batch 8 --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-verify" --filter "<5, 0>" --groups 7,8,10 > $O/c2_int_counters.txt 2> $O/c2_int_counters.err
tail -60 $O/c2_int_counters.txt
