// General control-cycle kernel: ONE WORKGROUP PER ROBOT INSTANCE, every matrix of the instance resident in LDS.
//
// Covers any serial chain with dof <= NMAX and any task stack (runtime sizes, batch-uniform control flow).
//   NMAX = 8,  T = 64  : one wavefront per instance (Panda-class arms), ~12 KB LDS -> up to 13 instances per CU
//   NMAX = 32, T = 256 : four wavefronts per instance (humanoid-scale chains), ~125 KB LDS -> 1 instance per CU
//
// What it computes (reference call stack):
//   SaiModel::updateModel()                       -> fk_prefix(), composite_inertia(), mass_matrix(), spd_inverse()
//   RobotController::updateControllerTaskModels() -> per task: projected Jacobian, range / singular values, Lambda, N
//       /root/reference/src/RobotController.cpp:68-77, tasks/MotionForceTask.cpp:247-268,
//       tasks/SingularityHandler.cpp:75-228 (non-singular branch), tasks/JointTask.cpp:218-283
//   RobotController::computeControlTorques()      -> per task control law + torque mapping + disturbance compensation
//       /root/reference/src/RobotController.cpp:79-118, tasks/MotionForceTask.cpp:270-509,
//       tasks/SingularityHandler.cpp:297-309, tasks/JointTask.cpp:285-356
//
// Algorithms differ from the reference's (and from the CPU oracle's) on purpose:
//   - forward kinematics as a log-step prefix product over the chain (all lanes busy) instead of a serial walk,
//   - M(q) from world-frame composite spatial inertias (suffix sums) instead of RBDL's CRBA recursion,
//   - every inverse is an SPD inverse via Cholesky (M, M_BIE, J M^-1 J^T) instead of general LU inverses,
//   - the SVDs of SingularityHandler / matrixRangeBasis are replaced by a parallel-ordered Jacobi eigen-solve of the
//     small Gram matrix Jp Jp^T: its eigenvalues are sigma^2 (branch predicates), its eigenvectors the left singular
//     vectors (range basis).  For the fully non-singular branch the torques are invariant to the choice of
//     orthonormal basis of range(Jp), so the basis of range(P) is used directly.
#include "saip_wg_cycle.h"

namespace saip {

template <int NMAX, int T>
__global__ void __launch_bounds__(T) saip_cycle_wg(const CycleParams P) {
	__shared__ WgSmem<NMAX> sm;
	const int b = blockIdx.x;
	if (b >= P.B) return;
	if (P.only_flagged && (P.status[b] & 1) == 0) return;  // status-driven slow path (diagnostic form; the engine uses the list form below)
	wg_cycle<NMAX, T>(P, b, sm);
}
// Device-side slow path behind the lane / eight-lane kernels: a fixed grid strides over the list of instances those kernels could not
// finish (CycleParams::flag_list).  No host round trip; when the list is empty every workgroup reads the count and leaves.
template <int NMAX, int T>
__global__ void __launch_bounds__(T) saip_cycle_wg_list(const CycleParams P) {
	__shared__ WgSmem<NMAX> sm;
	int count = *P.flag_count;  // wave-uniform scalar load; written by the kernel launched just before on the same stream
	count = count < P.B ? count : P.B;
	for (int i = blockIdx.x; i < count; i += gridDim.x) {
		const int b = P.flag_list[i];
		if (b >= 0 && b < P.B) wg_cycle<NMAX, T>(P, b, sm);
		__syncthreads();  // the next instance reuses the LDS block
	}
}

// ---------------------------------------------------------------- reInitializeTask: goal := current pose (one lane per instance)
__global__ void saip_reinit_kernel(const CycleParams P) {
	const int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= P.B) return;
	const ModelDev& md = *P.model;
	const int n = P.n;
	const int mask = P.reinit_mask;
	for (int t = 0; t < P.nt; t++) {
		if (P.reinit_task >= 0 && t != P.reinit_task) continue;
		const TaskDev& tk = P.tasks[t];
		if (tk.type == TASK_MOTION_FORCE) {
			double pw[3], Rc[9];
			fk_control_frame(md, tk, P.q, P.ld, b, pw, Rc);
			if (mask & 1) {  // linear part: goal position := current, velocity/acceleration := 0, integrator := 0
				for (int e = 0; e < 3; e++) {
					tk.goal[(size_t)e * P.ld + b] = pw[e];
					tk.goal[(size_t)(12 + e) * P.ld + b] = 0.0;
					tk.goal[(size_t)(18 + e) * P.ld + b] = 0.0;
					tk.integ[(size_t)e * P.ld + b] = 0.0;
					tk.integ[(size_t)(6 + e) * P.ld + b] = 0.0;  // force integrator (resetIntegratorsLinear)
				}
			}
			if (mask & 2) {  // angular part
				for (int e = 0; e < 9; e++) tk.goal[(size_t)(3 + e) * P.ld + b] = Rc[e];
				for (int e = 0; e < 3; e++) {
					tk.goal[(size_t)(15 + e) * P.ld + b] = 0.0;
					tk.goal[(size_t)(21 + e) * P.ld + b] = 0.0;
					tk.integ[(size_t)(3 + e) * P.ld + b] = 0.0;
					tk.integ[(size_t)(9 + e) * P.ld + b] = 0.0;  // moment integrator (resetIntegratorsAngular)
				}
			}
			if (mask & 4)  // goal force / moment (reInitializeTask, MotionForceTask.cpp:232-237)
				for (int e = 24; e < 36; e++) tk.goal[(size_t)e * P.ld + b] = 0.0;  // goal and sensed force / moment
		} else {
			for (int i = 0; i < tk.m; i++) {
				double cur = 0.0;
				for (int j = 0; j < n; j++) cur = fma(tk.S[i * n + j], P.q[(size_t)j * P.ld + b], cur);
				tk.goal[(size_t)i * P.ld + b] = cur;
				tk.goal[(size_t)(tk.m + i) * P.ld + b] = 0.0;
				tk.goal[(size_t)(2 * tk.m + i) * P.ld + b] = 0.0;
				tk.integ[(size_t)i * P.ld + b] = 0.0;
			}
		}
	}
}

// ---------------------------------------------------------------- host-callable launchers
hipError_t launch_cycle_wg_list(const CycleParams& P, hipStream_t stream) {
	// one workgroup per CU.  An empty list (the usual case) still costs a dependent launch: 4.4 - 4.6 us by rocprofv3 whatever the grid (4, 16, 64
	// or 256 workgroups measured alike, round 3: tools/ab_listgrid.sh) -- the price is the launch, not the 256 scalar loads
	const int grid = P.B < 256 ? P.B : 256;
	if (P.n <= 8) {
		hipLaunchKernelGGL((saip_cycle_wg_list<8, 64>), dim3(grid), dim3(64), 0, stream, P);
	} else {
		hipLaunchKernelGGL((saip_cycle_wg_list<32, 512>), dim3(grid < 256 ? grid : 256), dim3(512), 0, stream, P);
	}
	return hipGetLastError();
}
hipError_t launch_cycle_wg(const CycleParams& P, hipStream_t stream) {
	if (P.n <= 8) {
		hipLaunchKernelGGL((saip_cycle_wg<8, 64>), dim3(P.B), dim3(64), 0, stream, P);
	} else {
		// eight wavefronts per instance (two per SIMD: the LDS block allows one workgroup per CU, so the second wavefront of a SIMD is
		// the only latency hiding there is): 2.33 ms per 4096-instance launch of config 5 against 2.47 ms with four (SAIP_WG_T256=1)
		static const bool t256 = getenv("SAIP_WG_T256") != nullptr;
		if (t256) hipLaunchKernelGGL((saip_cycle_wg<32, 256>), dim3(P.B), dim3(256), 0, stream, P);
		else hipLaunchKernelGGL((saip_cycle_wg<32, 512>), dim3(P.B), dim3(512), 0, stream, P);
	}
	return hipGetLastError();
}
// current pose of the control frame of one motion-force task (MotionForceTask::getCurrentPosition / getCurrentOrientation): out = [12][ld],
// rows 0..2 position, 3..11 rotation (row-major)
__global__ void saip_pose_kernel(const CycleParams P, const int task, double* out) {
	const int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= P.B) return;
	double pw[3], Rc[9];
	fk_control_frame(*P.model, P.tasks[task], P.q, P.ld, b, pw, Rc);
	for (int e = 0; e < 3; e++) out[(size_t)e * P.ld + b] = pw[e];
	for (int e = 0; e < 9; e++) out[(size_t)(3 + e) * P.ld + b] = Rc[e];
}
hipError_t launch_pose(const CycleParams& P, int task, double* out, hipStream_t stream) {
	hipLaunchKernelGGL(saip_pose_kernel, dim3((P.B + 63) / 64), dim3(64), 0, stream, P, task, out);
	return hipGetLastError();
}

hipError_t launch_reinit(const CycleParams& P, hipStream_t stream) {
	hipLaunchKernelGGL(saip_reinit_kernel, dim3((P.B + 63) / 64), dim3(64), 0, stream, P);
	return hipGetLastError();
}

#if defined(SAIP_STAMP_WG)
// returns the number of log entries of the last launch and rewinds the log
extern "C" int saip_debug_stamps_wg(unsigned long long* out, int n) {
	int pos = 0, zero = 0;
	if (hipMemcpyFromSymbol(&pos, HIP_SYMBOL(saip_stamps_wg_pos), sizeof(int)) != hipSuccess) return -1;
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(saip_stamps_wg), sizeof(unsigned long long) * (n < 4096 ? n : 4096)) != hipSuccess) return -1;
	if (hipMemcpyToSymbol(HIP_SYMBOL(saip_stamps_wg_pos), &zero, sizeof(int)) != hipSuccess) return -1;
	return pos < 2047 ? pos : 2047;
}
#endif
}  // namespace saip
