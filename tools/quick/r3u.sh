#!/bin/bash
# validation of the compact quotient plane + everything since the last full run
set -u
export TMPDIR=/tmp
O=gpurun_out/r3u; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q -rs > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt); tail -4 $O/pytest.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-220 $O/bench.json
timeout 600 python bench.py --workload bfv_c4 --no-cpu-baseline --no-pmc | cut -c1-200
