// The large-batch (LEAN) instantiation of the lane kernel as its own translation unit, so that it can be compiled with the default
// machine scheduling strategy while the latency variants use max-ilp (capi.build_library).  All code is in saip_kernel_lane.hip.
#define SAIP_LANE_LEAN_TU 1
#undef SAIP_STAMP  // the diagnostic stamps belong to the latency variants' translation unit
#define SAIP_FENCE 1    // phase scheduling barriers and the scalar-cache prewarm still pay off for this instantiation
#define SAIP_PREWARM 1
#include "saip_kernel_lane.hip"
