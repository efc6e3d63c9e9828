#!/bin/bash
# build seal_amd/lib/variants/NAME.so = the product library with some kernel sources compiled with extra flags
# usage: tools/quick/build_variant.sh NAME "-DSEALHIP_KS_NT=15 ..." [FILES="ntt2_kernels.hip evaluator_keyswitch.cpp"]   (default: ntt2_kernels.hip)
# every listed file is also compiled with -DSEALHIP_AB_SWITCHES: its development environment switches (shl_ab_getenv) work in the variant
set -eu
NAME=$1; FLAGS=${2:-}; FILES=${3:-ntt2_kernels.hip}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
LIB=$ROOT/seal_amd/lib
make -s -j8 -C $ROOT/seal_amd/csrc gpu
mkdir -p $LIB/variants $LIB/obj_var
OBJS=$(ls $LIB/obj/*.o)
for f in $FILES; do
  b=${f%.*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DSEALHIP_AB_SWITCHES $FLAGS \
    -Rpass-analysis=kernel-resource-usage -x hip -c $ROOT/seal_amd/csrc/$f -o $LIB/obj_var/${b}_$NAME.o 2> $LIB/obj_var/${b}_$NAME.log || { tail -20 $LIB/obj_var/${b}_$NAME.log; exit 1; }
  OBJS=$(echo "$OBJS" | grep -v "/$b.o")" $LIB/obj_var/${b}_$NAME.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $LIB/variants/$NAME.so $OBJS -lz -ldl
grep -A12 "Function Name: .*ks2_kernelILi8ELi1E" $LIB/obj_var/ntt2_kernels_$NAME.log 2>/dev/null | grep -E "Name|VGPRs:|LDS Size|Occupancy|Spill" | head -8 || true
