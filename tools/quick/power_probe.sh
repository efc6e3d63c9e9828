#!/bin/bash
# GPU package power and shader clock sampled by rocm-smi while the headline step loops (and while the NTT leg loops): is the key switch
# power-limited?  (profiles/r05_inv_pb_fusion_bound.txt)   usage: tools/quick/power_probe.sh
export TMPDIR=/tmp
common="--no-cpu-baseline --no-pmc --no-verify --no-children"
rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk|mclk' | head -8
sample() { for i in $(seq 1 $1); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Package Power|sclk' | tr '\n' ' ' | sed 's/GPU\[0\]\s*: //g'; echo; sleep 0.25; done; }
echo "== idle"; sample 2
echo "== headline step loop (batch 256, 400 steps)"
python bench.py --steps 400 --warmup 2 $common > /tmp/pp_step.log 2>&1 &
P=$!; sleep 9; sample 14; wait $P; tail -1 /tmp/pp_step.log | cut -c1-160
echo "== NTT leg only"
python bench.py --ntt-only $common > /tmp/pp_ntt.log 2>&1 &
P=$!; sleep 3.5; sample 8; wait $P
