#!/bin/bash
# registers / scratch / LDS of every kernel of one translation unit, from the compiler's own remarks (no GPU needed):
#   tools/kernel_resources.sh csrc/saip_kernel_oct.hip [extra hipcc flags]
# (the cycle kernels are built with the max-ilp scheduling strategy, as capi.build_library does)
set -e
cd "$(dirname "$0")/../sai-primitives_amd"
src=$1; shift
sched=""
case "$src" in *kernel_lane.hip|*kernel_wg.hip|*kernel_oct.hip|*kernel_octjf.hip) sched="-mllvm -amdgpu-sched-strategy=max-ilp";; esac
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -x hip -c $sched -Rpass-analysis=kernel-resource-usage "$@" -o /tmp/kres_$$.o "$src" 2>/tmp/kres_$$.txt || { cat /tmp/kres_$$.txt; exit 1; }
grep -E " Name:|VGPRs:|AGPRs|ScratchSize|LDS Size" /tmp/kres_$$.txt | sed 's/.*remark: [^ ]* //; s/\[-Rpass.*//' | paste - - - - - | sed 's/ \+/ /g' | while read -r l; do n=$(echo "$l" | sed 's/Name: \([^ ]*\).*/\1/' | c++filt | sed 's/(saip::CycleParams)//; s/saip:://g; s/void //'); echo "$n |$(echo "$l" | sed 's/Name: [^ ]* //')"; done
rm -f /tmp/kres_$$.o /tmp/kres_$$.txt
