#!/bin/bash
# diagnostic: build a variant of the eight-lane kernel object with extra flags and link it with the product's other objects:
#   tools/oct_variant.sh NAME [-DFLAG ...]   ->  sai-primitives_amd/libsaip_var_NAME.so   (use through SAIP_LIB; never shipped)
set -e
cd "$(dirname "$0")/../sai-primitives_amd"
NAME=$1; shift
SRC=${SAIP_VARIANT_SRC:-csrc/saip_kernel_oct.hip}
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -x hip -c -mllvm -amdgpu-sched-strategy=max-ilp "$@" -o build/saip_kernel_oct_var_$NAME.o $SRC
OBJS=$(ls build/*.hip.o build/*.cpp.o | grep -v saip_kernel_oct.hip.o)
hipcc --offload-arch=gfx950 -fPIC -shared -o libsaip_var_$NAME.so $OBJS build/saip_kernel_oct_var_$NAME.o
echo built libsaip_var_$NAME.so
