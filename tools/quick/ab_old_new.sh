#!/bin/bash
# A/B on one box: seal_amd/lib/variants/$1.so ("old") against the built library ("new"), alternating; prints the headline rate,
# the NTT roofline leg and the configs[1] chains of each run
set -u
OLD=${1:-old}; ROUNDS=${2:-2}
OUT=gpurun_out/ab_$OLD; mkdir -p $OUT
cp seal_amd/lib/libsealhip.so /tmp/new.so
for r in $(seq $ROUNDS); do for v in old new; do
  if [ $v = old ]; then cp seal_amd/lib/variants/$OLD.so seal_amd/lib/libsealhip.so; else cp /tmp/new.so seal_amd/lib/libsealhip.so; fi
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-verify > $OUT/$v$r.json 2> $OUT/$v$r.err
  python - $v$r $OUT/$v$r.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = j["roofline"]; c = j["roofline_configs1"]["chains"]
    print("%-6s %8.1f ct/s %7.3f ms/step | NTT 2^16 %7.1f GB/s | configs[1] fwd %7.1f inv %7.1f | all-fp fwd %7.1f inv %7.1f" % (
        sys.argv[1], j["value"], j["ms_per_step"], r["achieved"], c[0]["forward"]["achieved"], c[0]["inverse"]["achieved"],
        c[1]["forward"]["achieved"], c[1]["inverse"]["achieved"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]).read()[-300:])
PY
done; done
cp /tmp/new.so seal_amd/lib/libsealhip.so
