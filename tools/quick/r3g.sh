#!/bin/bash
# round 3, seventh device session: the bench lines as the driver runs them (new roofline_step), C4 kernel table
set -u
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r3g; mkdir -p $O
cp seal_amd/lib/variants/ksA.so seal_amd/lib/libsealhip.so
( time timeout 900 python bench.py > $O/bench_headline.json 2> $O/bench_headline.err ) 2>&1 | grep real; cut -c1-1500 $O/bench_headline.json
( time timeout 900 python bench.py --workload bfv_c4 > $O/bench_c4.json 2> $O/bench_c4.err ) 2>&1 | grep real; tail -3 $O/bench_c4.err; cut -c1-600 $O/bench_c4.json
( time timeout 900 python bench.py --workload rotate_c5 > $O/bench_c5.json 2> $O/bench_c5.err ) 2>&1 | grep real; tail -3 $O/bench_c5.err; cut -c1-600 $O/bench_c5.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/prof4 -o c4 -- python $R/bench.py --workload bfv_c4 --total-batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-verify > $O/prof4.log 2>&1)
DB=$(find $O/prof4 -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $O/c4_kernel_stats.txt 2>&1; head -26 $O/c4_kernel_stats.txt
rm -rf $O/prof4
