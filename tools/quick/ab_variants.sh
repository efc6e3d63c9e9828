cp seal_amd/lib/libsealhip.so /tmp/def.so
for v in old new old new; do cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so; echo "== $v"; timeout 300 python tools/bench_configs.py --configs C2,C4 --no-cpu 2>&1 | tail -2 | cut -c1-200; done
cp /tmp/def.so seal_amd/lib/libsealhip.so
