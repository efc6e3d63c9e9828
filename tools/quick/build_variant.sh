#!/bin/bash
# build seal_amd/lib/variants/NAME.so = the product library with ntt2_kernels.hip compiled with extra flags
# usage: tools/quick/build_variant.sh NAME "-DSEALHIP_KS2_TWB_REGS=1 ..."
set -eu
NAME=$1; FLAGS=${2:-}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
LIB=$ROOT/seal_amd/lib
make -s -j8 -C $ROOT/seal_amd/csrc gpu
mkdir -p $LIB/variants $LIB/obj_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DSEALHIP_AB_SWITCHES $FLAGS \
  -Rpass-analysis=kernel-resource-usage -c $ROOT/seal_amd/csrc/ntt2_kernels.hip -o $LIB/obj_var/ntt2_$NAME.o 2> $LIB/obj_var/ntt2_$NAME.log || { tail -20 $LIB/obj_var/ntt2_$NAME.log; exit 1; }
OBJS=$(ls $LIB/obj/*.o | grep -v ntt2_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $LIB/variants/$NAME.so $OBJS $LIB/obj_var/ntt2_$NAME.o -lz -ldl
grep -A12 "Function Name: .*ks2_kernelILi8ELi1E" $LIB/obj_var/ntt2_$NAME.log | grep -E "Name|VGPRs:|LDS Size|Occupancy|Spill" | head -8
