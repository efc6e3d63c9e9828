#!/bin/bash
# headline step with the batch divided over S evaluators / streams (bench.py --streams S), same box
set -u
export TMPDIR=/tmp
for r in 1 2; do for S in 1 2 4; do
  hl=$(timeout 300 python bench.py --streams $S --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
  echo "streams $S (round $r) headline ct/s, ms/step: $hl"
done; done
