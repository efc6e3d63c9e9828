export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out/tl_rot; mkdir -p $O
for wl in rotate_c5 bfv_c4; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/prof_$wl -o t -- python $REPO/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-verify --no-children > $O/prof_$wl.log 2>&1)
DB=$(find $O/prof_$wl -name "*.db" | head -1)
python tools/step_timeline.py $DB --tail 420 > $O/timeline_$wl.txt 2>&1; rm -rf $O/prof_$wl
done
tail -3 $O/prof_rotate_c5.log
