#!/bin/bash
# A/B of library variants x env knobs: args "LABEL:LIBNAME:ENV=V,ENV=V" (LIBNAME = file in seal_amd/lib/variants or "default")
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cp seal_amd/lib/libsealhip.so /tmp/libsealhip_default.so
for spec in "$@"; do
  IFS=: read label libn envs <<< "$spec"
  if [ "$libn" = "default" ]; then cp /tmp/libsealhip_default.so seal_amd/lib/libsealhip.so; else cp seal_amd/lib/variants/$libn.so seal_amd/lib/libsealhip.so; fi
  envargs=$(echo "${envs:-X=1}" | tr ',' ' ')
  env $envargs timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/abl_$label.json 2> $OUT/abl_$label.err
  python - $label $OUT/abl_$label.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print("%-16s %8.1f ct/s  %6.3f ms/step   NTT %7.1f GB/s (%.4f ms)" % (sys.argv[1], j["value"], j["ms_per_step"], r["achieved"], r["ms_per_launch"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]).read()[-300:])
PY
done
cp /tmp/libsealhip_default.so seal_amd/lib/libsealhip.so
