#!/bin/bash
# does the 256 MiB Infinity Cache absorb the ks1 -> ks2 intermediate when it fits?  key-switch kernel times per ciphertext at batch 1 .. 64
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r3x; mkdir -p $O
for B in 1 2 4 8 64; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/prof$B -o hl -- python $R/bench.py --batch $B --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-verify > $O/prof$B.log 2>&1)
  DB=$(find $O/prof$B -name "*.db" | head -1)
  echo "== batch $B"; python tools/rocpd_summary.py $DB 2>&1 | grep -E "^kernel|ks1_kernel|ks2_kernel|tail2" | cut -c1-130
  rm -rf $O/prof$B
done 2>&1 | tee $O/batch_sweep.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_serialization.py -x -q -k "65536 or headline or key_save or lean or sampled" > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
