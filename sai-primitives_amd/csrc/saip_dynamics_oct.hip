// Forward dynamics + semi-implicit Euler for small batches of 7-dof chains: EIGHT LANES PER INSTANCE (see saip_kernel_oct.hip for
// the layout).  Same equations as saip_dynamics.hip,
//   M(q) qdd + b(q, dq) + g(q) = tau - damping dq ;  dq <- dq + dt qdd ;  q <- q + dt dq   (`substeps` times, torque held)
// the "step after the path" of SURVEY.md 8(f) f4 (/root/reference/examples/05-using_robot_controller/05-using_robot_controller.cpp:225-231),
// in a form where nothing walks the chain serially: with every spatial quantity expressed about the WORLD ORIGIN,
//   body velocity      V_i = sum_{j<=i} S_j dq_j                       (prefix sum over the lanes; S_j = (z_j, o_j x z_j) or (0, z_j))
//   bias acceleration  A_i = A_0 + sum_{j<=i} (V_j x S_j) dq_j          (prefix sum; A_0 = (0, -gravity) carries the gravity field)
//   body bias wrench   F_i = I_i A_i + V_i x* (I_i V_i)                  (per lane)
//   joint bias force   h_j = S_j . sum_{i>=j} F_i                        (suffix sum)
// and M(q) from composite inertias (suffix sum) exactly as in the cycle kernel; prefix / suffix sums are DPP scans.  M and the
// right-hand side are gathered through LDS, every lane factors M (Cholesky) and solves for the whole qdd, keeping its own entry.
#include "saip_dynamics_oct_body.h"

namespace saip {

__global__ void __launch_bounds__(64) saip_integrate_oct(const SimParams S) {
	__shared__ OctDynInst smem[8];
	integrate_oct_body(S, smem, blockIdx.x, threadIdx.x);
}

hipError_t launch_integrate_oct(const SimParams& S, hipStream_t stream) {
	const dim3 grid((S.B + 7) / 8), block(64);
	hipLaunchKernelGGL(saip_integrate_oct, grid, block, 0, stream, S);
	return hipGetLastError();
}

}  // namespace saip
