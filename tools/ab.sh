#!/bin/bash
# Same-box A/B of library variants and environment knobs (run on the GPU box: `gpurun -- tools/ab.sh [options] SPEC...`).
# One tool for what rounds 1-3 did with a one-off script per experiment (tools/quick/r3*.sh, ab_*.sh: removed in round 4; the
# experiments themselves are recorded in profiles/r0?_*experiments.txt).
#
#   SPEC = LABEL:LIB[:ENV=V,ENV=V...]     LIB = "default" or a file NAME in seal_amd/lib/variants/NAME.so
#                                         (built here with tools/quick/build_variant.sh NAME "-DFLAG ...")
#   options (before the specs):
#     --rounds R          repeat the whole list R times, interleaved (default 2: box drift shows as the spread of a label)
#     --workload W        headline (default) | bfv_c4 | rotate_c5 | c2 (configs[1] NTT chain via tools/bench_configs.py) | ntt (bench.py --ntt-only)
#     --trace             additionally one rocprofv3 --kernel-trace pass per spec: the step's time line (tools/step_timeline.py)
#     --check "PYTEST -k EXPR"   run the GPU parity tests selected by EXPR with every non-default LIB first (a variant that breaks words is not measured)
#     --out DIR           default gpurun_out/ab
#     --bench-args "..."  extra bench.py arguments (e.g. "--batch 64")
set -u
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/ab; ROUNDS=2; WORKLOAD=headline; TRACE=0; CHECK=""; EXTRA=""
while [ $# -gt 0 ]; do
  case $1 in
    --rounds) ROUNDS=$2; shift 2 ;;
    --workload) WORKLOAD=$2; shift 2 ;;
    --trace) TRACE=1; shift ;;
    --check) CHECK=$2; shift 2 ;;
    --out) OUT=$REPO/$2; shift 2 ;;
    --bench-args) EXTRA=$2; shift 2 ;;
    *) break ;;
  esac
done
mkdir -p $OUT
cp seal_amd/lib/libsealhip.so /tmp/libsealhip_default.so
trap 'cp /tmp/libsealhip_default.so $REPO/seal_amd/lib/libsealhip.so' EXIT
use_lib() { if [ "$1" = "default" ]; then cp /tmp/libsealhip_default.so seal_amd/lib/libsealhip.so; else cp seal_amd/lib/variants/$1.so seal_amd/lib/libsealhip.so; fi; }
common="--no-cpu-baseline --no-pmc --no-verify --no-children"
one() {  # label lib envs round
  local label=$1 libn=$2 envs=$3 r=$4 envargs
  envargs=$(echo "${envs:-SEALHIP_AB_NONE=1}" | tr ',' ' ')
  use_lib $libn
  case $WORKLOAD in
    c2)  env $envargs timeout 300 python tools/bench_configs.py --configs C2 --no-cpu 2>/dev/null | grep fwd_GBs | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('%-18s round $r  configs[1] chain forward %.1f GB/s inverse %.1f GB/s' % ('$label', j['fwd_GBs'], j['inv_GBs']))" ;;
    ntt) env $envargs timeout 300 python bench.py --ntt-only $common $EXTRA 2>$OUT/$label.err | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read())['roofline']; print('%-18s round $r  NTT %.1f GB/s (%.4f ms per launch)' % ('$label', j['achieved'], j['ms_per_launch']))" ;;
    *)   env $envargs timeout 600 python bench.py --workload $WORKLOAD --steps 6 --warmup 2 $common $EXTRA 2>$OUT/$label.err | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j.get('roofline') or {}; print('%-18s round $r  %9.1f ct/s  %7.3f ms/step   NTT leg %s GB/s' % ('$label', j['value'], j['ms_per_step'], r.get('achieved')))" ;;
  esac
}
if [ -n "$CHECK" ]; then
  for spec in "$@"; do IFS=: read label libn envs <<< "$spec"
    [ "$libn" = "default" ] && continue
    use_lib $libn; envargs=$(echo "${envs:-SEALHIP_AB_NONE=1}" | tr ',' ' ')
    env $envargs timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "$CHECK" > $OUT/check_$label.txt 2>&1
    echo "check $label: $(grep -E 'passed|failed|error' $OUT/check_$label.txt | tail -1)"
  done
fi
for r in $(seq 1 $ROUNDS); do for spec in "$@"; do IFS=: read label libn envs <<< "$spec"; one "$label" "$libn" "${envs:-}" $r; done; done 2>&1 | tee $OUT/ab.txt
if [ $TRACE = 1 ]; then
  for spec in "$@"; do IFS=: read label libn envs <<< "$spec"
    use_lib $libn; envargs=$(echo "${envs:-SEALHIP_AB_NONE=1}" | tr ',' ' ')
    wl=$WORKLOAD; case $wl in c2|ntt) wl=headline ;; esac
    (cd /tmp && env $envargs timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_$label -o t -- python $REPO/bench.py --workload $wl --steps 3 --warmup 1 $common $EXTRA > $OUT/prof_$label.log 2>&1)
    DB=$(find $OUT/prof_$label -name "*.db" | head -1)
    python tools/step_timeline.py $DB > $OUT/timeline_$label.txt 2>&1; rm -rf $OUT/prof_$label
    echo "== $label"; tail -24 $OUT/timeline_$label.txt
  done
fi
