#!/bin/bash
# bench.py's own configs[1] measurement (4096 polynomials) under environment variants: tools/quick/configs1_ab.sh ROUNDS LABEL=ENV ...
set -u
export TMPDIR=/tmp
R=$1; shift
for r in $(seq 1 $R); do for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=SEALHIP_AB_NONE=1
  env $(echo $envs | tr ',' ' ') python bench.py --ntt-only --no-cpu-baseline --no-pmc --no-verify --no-children 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read())['roofline_configs1']['chains']
print('%-10s round $r  ' % '$label' + '   '.join('%s fwd %.4f inv %.4f' % (c['chain'][:14], c['forward']['frac'], c['inverse']['frac']) for c in j))"
done; done
