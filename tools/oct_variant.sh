#!/bin/bash
# diagnostic: build a variant of ONE kernel object with extra flags and link it with the product's other objects (never shipped):
#   tools/oct_variant.sh NAME [-DFLAG ...]                      -> sai-primitives_amd/libsaip_var_NAME.so   (use through SAIP_LIB)
#   SAIP_VARIANT_KERNEL=wg tools/oct_variant.sh NAME -DSAIP_STAMP_WG   (the general kernel instead of the eight-lane one)
#   SAIP_VARIANT_SRC=csrc/other.hip: compile that source in place of the kernel's own;  SAIP_VARIANT_SCHED=default|max-ilp|...: machine scheduler strategy
set -e
cd "$(dirname "$0")/../sai-primitives_amd"
NAME=$1; shift
K=${SAIP_VARIANT_KERNEL:-oct}
SRC=${SAIP_VARIANT_SRC:-csrc/saip_kernel_$K.hip}
SCHED=${SAIP_VARIANT_SCHED:-max-ilp}
if [ "$SCHED" = "default" ]; then SFLAGS=""; else SFLAGS="-mllvm -amdgpu-sched-strategy=$SCHED"; fi
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -x hip -c $SFLAGS "$@" -o build/saip_kernel_${K}_var_$NAME.o $SRC
OBJS=$(ls build/*.hip.o build/*.cpp.o | grep -v saip_kernel_$K.hip.o)
hipcc --offload-arch=gfx950 -fPIC -shared -o libsaip_var_$NAME.so $OBJS build/saip_kernel_${K}_var_$NAME.o
echo built libsaip_var_$NAME.so
