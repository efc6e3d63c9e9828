#!/bin/bash
# A/B of runtime knobs: each argument is "LABEL:ENV1=V1,ENV2=V2" ; runs bench.py --ntt-only (fast) or the full bench with FULL=1
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  envargs=$(echo "$envs" | tr ',' ' ')
  if [ "${FULL:-0}" = "1" ]; then extra="--steps 6 --warmup 2 --no-cpu-baseline"; else extra="--ntt-only"; fi
  env $envargs timeout 300 python bench.py $extra > $OUT/abe_$label.json 2> $OUT/abe_$label.err
  python - $label $OUT/abe_$label.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = j.get("roofline", j)
    print("%-14s %8.1f ct/s  %6.3f ms/step   NTT %7.1f GB/s (%.4f ms)" % (sys.argv[1], j.get("value") or 0, j.get("ms_per_step") or 0, r["achieved"], r["ms_per_launch"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]).read()[-300:])
PY
done
