// Control-cycle kernel for SMALL batches of 7-dof arms: EIGHT LANES PER ROBOT INSTANCE (eight instances per wavefront).
//
// At the batch the headline metric is quoted on (4096 instances per GPU) the lane-per-instance kernel fills 64 of the 1024 SIMDs
// of an MI355X and each of those wavefronts walks the whole 6 300-instruction cycle alone.  Here the rows of every matrix of
// an instance are spread over eight lanes (lane r owns joint r / row r; lane 7 is padding), operands the other lanes need travel
// by DPP (scans along the chain, reductions) or through a small LDS block per instance, and the inherently serial factorisations
// are done redundantly (a Cholesky factor per lane, then each lane solves for its own column of the inverse).  4096 instances
// become 512 wavefronts of 2 500 instructions each -- and, while the launch fits the chip in one round, 1024: a second wavefront per
// group of eight instances (oct_cycle_body<..., ROLE>) runs what does not need the mass matrix next to the inertia
// chain.  Synchronisation inside a wavefront is a wavefront-scope fence (oct_sync); the two wavefronts of a group meet at two or three
// workgroup barriers.
//
// Covers { MotionForceTask, JointTask } stacks on a 7-dof chain (engine: oct_eligible): the headline stack of BASELINE config 2 (full
// 6-dof task + full joint task; <false, 0> is its lean instantiation), partial motion-force tasks with a joint task of <= 4 rows
// (config 3, GJ = 1) or a full joint task (GJ = 2), default or general control laws (no closed-loop force control), joint limit
// avoidance, gravity compensation, torque saturation, disableSingularityHandling() (near-singular instances reduced in the kernel)
// and, for the headline stack, SingularityHandler's blended type-1 / type-2 strategies with their per-instance state.  Instances
// outside the non-singular branch of the other stacks are flagged and, with the blended strategies on, recomputed by the general
// kernel launched behind over the flag list.
//   SaiModel::updateModel()                       -> prefix-product kinematics, composite inertias, M, M^-1, M_BIE^-1
//   RobotController::updateControllerTaskModels() -> /root/reference/src/RobotController.cpp:68-77, tasks/MotionForceTask.cpp:247-268,
//       tasks/SingularityHandler.cpp:75-228 (non-singular branch; others flagged -> slow path), tasks/JointTask.cpp:218-283
//   RobotController::computeControlTorques()      -> RobotController.cpp:79-118, MotionForceTask.cpp:270-509,
//       SingularityHandler.cpp:297-309, JointTask.cpp:285-356
#include <hip/hip_runtime.h>
#include <math.h>

#include "saip_device.h"
#include "saip_dynamics_oct_body.h"
#include "saip_fk.h"
#include "saip_law.h"
#include "saip_oct_common.h"
// the general kernel's body as this kernel's slow tail: one wavefront plays the 64-thread workgroup, so its synchronisation points are
// wavefront-local (the other wavefront of a two-wavefront workgroup has left by then)
#define SAIP_WG_SYNC()                                          \
	do {                                                        \
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
		__builtin_amdgcn_wave_barrier();                        \
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
	} while (0)
#include "saip_wg_cycle.h"

namespace saip {

#if defined(SAIP_STAMP_OCT)
// diagnostic build only (never shipped): s_memtime of wavefront 0 at the phase boundaries
__device__ unsigned long long saip_stamps_oct[1024 * 64];  // [workgroup][wavefront][stamp]
#define STAMP(i)                                                                              \
	do {                                                                                      \
		__builtin_amdgcn_sched_barrier(0);                                                    \
		if (blockIdx.x < 1024 && (threadIdx.x & 63) == 0) saip_stamps_oct[blockIdx.x * 64 + (threadIdx.x >> 6) * 32 + (i)] = __builtin_amdgcn_s_memtime(); \
		__builtin_amdgcn_sched_barrier(0);                                                    \
	} while (0)
#else
#define STAMP(i)
#endif

struct OctInst {
	union {  // the kinematics scratch is dead before the task matrices are written
		double X[8][12];     // world frame per joint: R (9, row-major) + o (3)
		struct {
			double T1[6][8], A[8][8], Am[8][8], Lam[8][8];  // Lam also carries J M_BIE^-1 until Lambda exists; rows 6, 7: padding lanes' scratch
			double N1[8][8];
		};
	};
	double M[8][8];      // lower triangle of M(q), row r by lane r; kept to the end (the blended singularity strategies want M z)
	double zo[8][6];     // joint motion vector about the world origin S_r = (w, v): revolute (z, o x z), prismatic (0, z)
	double J[6][8];
	double vec[13][8];   // 0 dq, 1 tau, 2 g, 3 / 4 row and scalar exchanges, 5 goal force + moment (general laws), 7 u, 8 d, 9 flags, 10..12 motion-force goal (24)
	double ist[10];      // integrator state, fetched with the inputs: 0..5 motion-force task (position, orientation), 6..9 the rows of a Gram-path joint
	                     // task; the control law leaves the advanced motion-force values here for the epilogue.  (Also the padding that makes the
	                     // instance stride = 2 (mod 32) doubles: the eight instances of a wavefront hit different LDS banks.)
};
static_assert(sizeof(OctInst) % 256 == 16, "instance stride must be 2 (mod 32) doubles");
static_assert(sizeof(OctInst) * 8 * 4 <= 160 * 1024, "four wavefronts per CU (one LDS block each, also with two wavefronts per instance group)");

// eigen-decomposition of a symmetric R x R matrix (R = 4, 6) held, identically, by all eight lanes of an instance: round-robin Jacobi.
// The R/2 rotations of a round touch disjoint index pairs, so they are computed at the same time in different lanes (lane m of each
// quad takes pair m; the reciprocal / rsqrt chains are most of a rotation's cost), broadcast inside the quad by DPP quad_perm and applied
// by every lane to its copy of A; the eigenvector matrix is distributed (lane i < R keeps row i: 4 instead of 4 R operations per
// rotation) and gathered at the end.  Wave-uniform convergence test per sweep.  Eigenvalues on the diagonal of A, eigenvectors in the
// columns of V (all lanes alike).  (The row-cyclic form this replaces computed every rotation in every lane: 92 instructions per pair
// against ~49 now; config 14: 37 k clocks of the blended block's 100 k.)
template <int R>
__device__ __forceinline__ void oct_jacobi_n(double (&A)[R][R], double (&V)[R][R], const int lane) {
	static_assert(R == 4 || R == 6, "even sizes with at most three pairs per round");
	constexpr int NP = R / 2, RM = R - 1;
	// which pair of a round this lane computes.  Side by side: lane m of each quad takes pair m (quad_perm broadcasts 0x00 / 0x55 / 0xAA).
	// Interleaved: a quad holds joints 2 q, 2 q + 1 of two instances -- even joints take pair 0, odd joints pair 1 (quad_perm [0,1,0,1] /
	// [2,3,2,3] hands them to the quad's lanes of the same instance), and the third pair of a 6 x 6 round is computed by every lane itself
	// (fetching all three by ds_bpermute from joints 0, 1, 2 measured the same or slower: config 12 27.4 vs 26.7 us; side by side 25.0).
	const int r = octl_r(lane), m = OCT_ILV ? (r & 1) : ((lane & 3) < NP ? (lane & 3) : 0);
	// t = tan of the rotation angle: sign(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (aqq - app) / (2 apq), written without the
	// division by apq; hardware reciprocal / rsqrt estimates + Newton steps (arguments are positive and in range)
	auto rotation = [](const double apq, const double app, const double aqq, double& cs, double& sn, double& tn) {
		const double d = aqq - app, b2 = 2.0 * apq;
		const double x = fma(d, d, b2 * b2);
		const bool rot = x > 1e-300;
		const double h = rot ? x * oct_rsqrt(x) : 1.0;
		const double tm = fabs(b2) * oct_rcp(fabs(d) + h);
		tn = rot ? (((d >= 0.0) == (b2 >= 0.0)) ? tm : -tm) : 0.0;
		cs = oct_rsqrt(fma(tn, tn, 1.0));
		sn = tn * cs;
	};
	double vrow[R];
	UNR for (int j = 0; j < R; j++) vrow[j] = (j == r) ? 1.0 : 0.0;
	for (int sweep = 0; sweep < 12; sweep++) {
		double off = 0.0, dg = 0.0;
		UNR for (int p = 0; p < R; p++) {
			dg = fma(A[p][p], A[p][p], dg);
			UNR for (int q = p + 1; q < R; q++) off = fma(A[p][q], A[p][q], off);
		}
		if (!__any(off > 1e-32 * dg)) break;  // every instance of the wavefront has converged (quadratic convergence: 3 - 6 sweeps)
		UNR for (int t = 0; t < RM; t++) {
			// tournament round t: (t, R-1) and ((t + j) mod (R-1), (t - j) mod (R-1)), j = 1 .. R/2 - 1
			double apq = 0.0, app = 0.0, aqq = 0.0;
			UNR for (int j = 0; j < NP; j++) {
				const int a_ = j == 0 ? t : (t + j) % RM, b_ = j == 0 ? RM : (t + RM - j) % RM;
				const int p = a_ < b_ ? a_ : b_, q = a_ < b_ ? b_ : a_;
				const bool mine = m == j;
				apq = mine ? A[p][q] : apq;
				app = mine ? A[p][p] : app;
				aqq = mine ? A[q][q] : aqq;
			}
			double cs, sn, tn, cs2 = 1.0, sn2 = 0.0, tn2 = 0.0;
			rotation(apq, app, aqq, cs, sn, tn);
			if (OCT_ILV && NP == 3) {  // the third pair of the round, in every lane
				const int a_ = (t + 2) % RM, b_ = (t + RM - 2) % RM;
				const int p = a_ < b_ ? a_ : b_, q = a_ < b_ ? b_ : a_;
				rotation(A[p][q], A[p][p], A[q][q], cs2, sn2, tn2);
			}
			UNR for (int j = 0; j < NP; j++) {
				const int a_ = j == 0 ? t : (t + j) % RM, b_ = j == 0 ? RM : (t + RM - j) % RM;
				const int p = a_ < b_ ? a_ : b_, q = a_ < b_ ? b_ : a_;
				double c, s_, tj;
				if (OCT_ILV) {
					if (j == 0) { c = oct_dpp<0x44>(cs); s_ = oct_dpp<0x44>(sn); tj = oct_dpp<0x44>(tn); }
					else if (j == 1) { c = oct_dpp<0xEE>(cs); s_ = oct_dpp<0xEE>(sn); tj = oct_dpp<0xEE>(tn); }
					else { c = cs2; s_ = sn2; tj = tn2; }
				} else if (j == 0) { c = oct_dpp<0x00>(cs); s_ = oct_dpp<0x00>(sn); tj = oct_dpp<0x00>(tn); }
				else if (j == 1) { c = oct_dpp<0x55>(cs); s_ = oct_dpp<0x55>(sn); tj = oct_dpp<0x55>(tn); }
				else { c = oct_dpp<0xAA>(cs); s_ = oct_dpp<0xAA>(sn); tj = oct_dpp<0xAA>(tn); }
				const double apq_j = A[p][q];
				UNR for (int k = 0; k < R; k++) {
					if (k != p && k != q) {
						const double akp = A[k][p], akq = A[k][q];
						const double np_ = c * akp - s_ * akq, nq_ = s_ * akp + c * akq;
						A[k][p] = np_;
						A[p][k] = np_;
						A[k][q] = nq_;
						A[q][k] = nq_;
					}
				}
				A[p][p] = fma(-tj, apq_j, A[p][p]);
				A[q][q] = fma(tj, apq_j, A[q][q]);
				A[p][q] = 0.0;
				A[q][p] = 0.0;
				const double vp = vrow[p], vq = vrow[q];
				vrow[p] = c * vp - s_ * vq;
				vrow[q] = s_ * vp + c * vq;
			}
		}
	}
	UNR for (int i = 0; i < R; i++)
		UNR for (int c = 0; c < R; c++) V[i][c] = __shfl(vrow[c], octl_src(lane, i));
}
__device__ __forceinline__ void oct_jacobi4(double (&A)[4][4], double (&V)[4][4], const int lane) { oct_jacobi_n<4>(A, V, lane); }
__device__ __forceinline__ void oct_spd_inverse4(const double (&A)[4][4], double (&Ainv)[4][4]) { oct_spd_inverse_n<4>(A, Ainv); }


// world frame of this lane's joint for posture q (one joint angle per lane of the instance): local transform, then the three-step
// prefix product over the chain by DPP row_shr (the same arithmetic as the kernel's own kinematics; used by the singularity
// classification, which looks at the control frame of two perturbed postures per singular direction)
__device__ __forceinline__ void oct_fk_frame(const ModelDev& md, const int rr, const int r, const double q, double (&Tw)[12]) {
	const double* R0 = md.R0[rr];
	const double ax = md.axis[rr][0], ay = md.axis[rr][1], az = md.axis[rr][2];
	if (md.jtype[rr] == 1) {
		double s, c;
		sincos_joint(q, &s, &c);
		const double v = 1.0 - c;
		const double Rq[9] = {c + ax * ax * v,      ax * ay * v - az * s, ax * az * v + ay * s,
							  ay * ax * v + az * s, c + ay * ay * v,      ay * az * v - ax * s,
							  az * ax * v - ay * s, az * ay * v + ax * s, c + az * az * v};
		UNR for (int i = 0; i < 3; i++)
			UNR for (int j = 0; j < 3; j++) Tw[3 * i + j] = R0[3 * i] * Rq[j] + R0[3 * i + 1] * Rq[3 + j] + R0[3 * i + 2] * Rq[6 + j];
		UNR for (int e = 0; e < 3; e++) Tw[9 + e] = md.p0[rr][e];
	} else {
		UNR for (int e = 0; e < 9; e++) Tw[e] = R0[e];
		const double a[3] = {ax, ay, az};
		double d[3];
		oct_mat3_vec(R0, a, d);
		UNR for (int e = 0; e < 3; e++) Tw[9 + e] = md.p0[rr][e] + d[e] * q;
	}
	OCT_FRAME_STEP(1, Tw, r)
	OCT_FRAME_STEP(2, Tw, r)
	OCT_FRAME_STEP(4, Tw, r)
}


// FULL: general control laws, partial motion-force tasks and the joint-limit-avoidance wrap compiled in; GJ: general joint task (<= 4 rows); the lean instantiation is the benchmarked default stack
// (their mere presence costs it 1.7 % through register allocation)
// DUO (the lean instantiation only): TWO wavefronts per group of eight instances, on two SIMDs of the CU.  Wavefront A (0) runs the inertia
// chain -- composite inertias, M(q), its factor, M^-1, T1, A, the factor of A -- and then the tail; wavefront B (1) runs, at the same
// time, what does not need M: the branch predicate on G = J J^T and the motion-force control law, hands (singular, F) over through LDS
// at the one workgroup barrier of the kernel and exits.  Each has its own LDS block; every other synchronisation point is
// wavefront-local (LDS operations of one wavefront execute in order).
template <bool DUO>
__device__ __forceinline__ void oct_sync() {
	if (DUO) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	} else {
		__syncthreads();
	}
}

// ROLE 0: the whole cycle in one wavefront; 1 / 2: wavefront A / B of the two-wavefront form (compile-time, so that neither carries the
// other's registers: as run-time branches of one body the merged live ranges spilled)
// NN: dof of the chain.  7 everywhere except the general instantiation <true, 2, ...> (full joint task behind a motion-force task, rank
// n - k by multi-pivot Gram-Schmidt), which also exists for 6 and 8 dof: with 8 no lane is padding, with 6 two are.
template <bool FULL, int GJ, int ROLE, int NN = 7>
__device__ __forceinline__ bool oct_cycle_body(const CycleParams& P, OctInst* smem, OctInst* smemA, OctInst* smemB, int* wgflags = nullptr, double* lawc = nullptr) {
	constexpr bool DUO = ROLE != 0;
	constexpr bool roleA = ROLE != 2, roleB = ROLE != 1;
	constexpr int N = NN;
#if !defined(SAIP_OCT_NO_BROWS)
	// Lean two-wavefront form: B does not wait where A takes the Jacobian from it -- A polls a flag in LDS instead of both meeting at a workgroup
	// barrier (B was there 1.3 k clocks early) -- and B uses the time it gains to form the rows of A = T1 J^T from the T1 that A publishes
	// (second flag), which takes the second half of the T1 / A phase off A's chain: 7.04 -> 6.91 us (same session; -DSAIP_OCT_NO_BROWS for the A/B).
	// The flags live in LDS, are zeroed behind one early workgroup barrier, and every poll loop is bounded.
	constexpr bool BROWS = DUO && !FULL && GJ == 0;
	auto flag_wait = [&](const int k) {
		int it = 0, v = 0;
		while ((v = __hip_atomic_load(wgflags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0 && ++it < 200000) __builtin_amdgcn_s_sleep(1);
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		return v;
	};
	auto flag_set = [&](const int k, const int v) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		if ((threadIdx.x & 63) == 0) __hip_atomic_store(wgflags + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	};
#else
	constexpr bool BROWS = false;
#endif
	bool lost = false;  // a bounded poll ran out (cannot happen while both wavefronts of the workgroup run; if it ever does, the instances end flagged, not silently wrong)
	static_assert(NN == 7 || (FULL && GJ == 2), "6- and 8-dof chains: the general instantiation only");
	const int lane = threadIdx.x & 63;
	const int r = octl_r(lane), grp = octl_grp(lane);
	const int rr = r < N ? r : N - 1;  // padding lane mirrors joint 6 and never writes
	const bool act = r < N;
	OctInst& sm = smem[grp];
	OctInst& smA = smemA[grp];  // wavefront A's block (B reads A = J M^-1 J^T, T1 and M^-1 from it after the first barrier)
	OctInst& smB = smemB[grp];  // where wavefront B leaves its results
	const int b_raw = (int)blockIdx.x * 8 + grp;
	const bool live = b_raw < P.B;
	const int b = live ? b_raw : P.B - 1;
	const size_t ld = P.ld;
	const ModelDev& md = *P.model;
	const TaskDev& mf = P.tasks[0];
	const TaskDev& jt = P.tasks[1];
	const int kmf = FULL ? mf.k : 6;  // rank of the motion-force task (partial tasks: rows >= k of its Jacobian are zero, diagonals padded with 1)
	// the decoupling types decide branches in the tail of the cycle; read there (vector loads of a uniform word behind the fences) each cost a
	// memory round trip on the critical path -- requested here, made scalar behind the local transform
	const int mf_dec_v = mf.decoupling, jt_dec_v = jt.decoupling;
	// disableSingularityHandling(): near-singular tasks are reduced in the kernel instead of being flagged (batch-uniform)
	const bool trunc_mode = FULL && GJ == 2 && P.oct_truncate;

	STAMP(0);
	// ---------------------------------------------------------------- inputs (requested up front; consumed much later)
	const double q_r = P.q[(size_t)rr * ld + b];
	const double dq_r = P.dq[(size_t)rr * ld + b];
	// this joint's constants, requested together with the state.  (Left where they are used, the loads of R0 / p0 / axis sat INSIDE the branch on the
	// joint type -- a second memory round trip behind the one that fetched the type -- and those of the body's inertia behind the frames: two
	// exposed round trips at the start of every wavefront.)
	const int jtype_r = md.jtype[rr];
	double R0v[9], p0v[3], axv[3];
	UNR for (int e = 0; e < 9; e++) R0v[e] = md.R0[rr][e];
	UNR for (int e = 0; e < 3; e++) {
		p0v[e] = md.p0[rr][e];
		axv[e] = md.axis[rr][e];
	}
	double comv[3] = {0, 0, 0}, inev[6] = {0, 0, 0, 0, 0, 0}, mass_r = 0.0;
	if (ROLE == 0) {  // (the two-wavefront form asks for the body's inertia behind the local transform: more requests in front of the first wait made it slower)
		UNR for (int e = 0; e < 3; e++) comv[e] = md.com[rr][e];
		UNR for (int e = 0; e < 6; e++) inev[e] = md.inertia[rr][e];
		mass_r = md.mass[rr];
	}
	double gl0 = 0.0, gl1 = 0.0, gl2 = 0.0;  // 24 goal rows, 3 per lane
	if (roleB) {
		gl0 = saip_ldg(P.goal2[0], (size_t)r * ld + b);
		gl1 = saip_ldg(P.goal2[0], (size_t)(8 + r) * ld + b);
		gl2 = saip_ldg(P.goal2[0], (size_t)(16 + r) * ld + b);
	}
	const double gl3 = (roleB && FULL && mf.general_law && r < 6) ? saip_ldg(P.goal2[0], (size_t)(24 + r) * ld + b) : 0.0;  // goal force / moment of the general laws
	// partial motion-force tasks: the control law multiplies by ninety batch-uniform constants (Bm, the projections, the sigma matrices, the
	// gains: one contiguous run of the task block).  Read where the law runs -- behind a dozen wavefront fences -- they were 120 loads with
	// their memory round trip in front of the law (3.0 k of B's 12.3 k clocks on config 3); requested here, two per lane, parked in LDS.
	static_assert(offsetof(TaskDev, Ppos) == offsetof(TaskDev, Bm) + 36 * 8 && offsetof(TaskDev, Pori) == offsetof(TaskDev, Bm) + 45 * 8 &&
					  offsetof(TaskDev, sig_p) == offsetof(TaskDev, Bm) + 54 * 8 && offsetof(TaskDev, sig_o) == offsetof(TaskDev, Bm) + 63 * 8 &&
					  offsetof(TaskDev, kp_pos) == offsetof(TaskDev, Bm) + 72 * 8 && offsetof(TaskDev, ki_ori) == offsetof(TaskDev, Bm) + 87 * 8,
				  "the law constants are one contiguous run");
	double lc0 = 0.0, lc1 = 0.0;
	if (FULL && roleB) {
		const double* lsrc = reinterpret_cast<const double*>(reinterpret_cast<const char*>(&mf) + offsetof(TaskDev, Bm));
		lc0 = lsrc[lane];
		lc1 = lsrc[lane < 26 ? 64 + lane : 89];
	}
	const int mj = GJ == 1 ? jt.m : N;                 // rows of the joint task (Gram path: <= 4)
	const int rj = GJ == 1 ? (r < mj ? r : 0) : rr;    // the task row whose goal this lane fetches
	double jg_q = 0.0, jg_dq = 0.0, jg_ddq = 0.0;
	if (roleA) {
		jg_q = saip_ldg(P.goal2[1], (size_t)rj * ld + b);
		jg_dq = saip_ldg(P.goal2[1], (size_t)(mj + rj) * ld + b);
		jg_ddq = saip_ldg(P.goal2[1], (size_t)(2 * mj + rj) * ld + b);
	}

	// integrator state and this lane's joint gains: requested early, committed from registers / LDS in the epilogue (the loads used to sit in front
	// of the control laws and the commit re-read a shadow copy from HBM: 3.1 us per launch with the integrators on, measured; now 0.6).  WHERE
	// they are requested is measured too: the batch-uniform flags that decide about them are scalar loads from the task block, and asked for at
	// the very start they make the wavefront wait before it has issued anything else -- the two-wavefront form asks behind the local transform
	// (7.23 -> 7.00 us), the one-wavefront form, which has the motion-force goal loads in flight there anyway, at the start (9.18 vs 9.01 us).
#define OCT_REQUEST_STATE(SFX)                                                                                          \
	const bool track_mf##SFX = (mf.has_ki || P.integ_always), track_jt##SFX = (jt.has_ki || P.integ_always);            \
	double mfi_pre##SFX = 0.0, jti_pre##SFX = 0.0;                                                                      \
	if (roleB && track_mf##SFX && r < 6) mfi_pre##SFX = saip_ldg(P.integ2[0], (size_t)r * ld + b);                                   \
	if (roleA && track_jt##SFX) jti_pre##SFX = saip_ldg(P.integ2[1], (size_t)(GJ == 1 ? rj : rr) * ld + b);                          \
	double jkp_r##SFX = 0.0, jkv_r##SFX = 0.0, jki_r##SFX = 0.0; /* (per-lane gain loads: used by the control law behind M) */ \
	if (roleA && GJ != 1) {                                                                                             \
		jkp_r##SFX = jt.kp[rr];                                                                                         \
		jkv_r##SFX = jt.kv[rr];                                                                                         \
		jki_r##SFX = jt.ki[rr];                                                                                         \
	}
	double jt_ie_new = 0.0;
	OCT_REQUEST_STATE(_e)  // (dead code in the two-wavefront form: nothing below uses the _e values there)

	STAMP(1);
	// ---------------------------------------------------------------- updateModel(): kinematics (log-step prefix product over the chain)
	double Tw[12];
	{
		const double* R0 = R0v;
		const double ax = axv[0], ay = axv[1], az = axv[2];
		if (jtype_r == 1) {
			double s, c;
			sincos_joint(q_r, &s, &c);
			if (md.all_axis_z) {  // (batch-uniform) rotation about the local z axis: R0 Rz(q) mixes the first two columns of R0 only
				UNR for (int i = 0; i < 3; i++) {
					Tw[3 * i] = R0[3 * i] * c + R0[3 * i + 1] * s;
					Tw[3 * i + 1] = R0[3 * i + 1] * c - R0[3 * i] * s;
					Tw[3 * i + 2] = R0[3 * i + 2];
				}
			} else {
				const double v = 1.0 - c;
				const double Rq[9] = {c + ax * ax * v,      ax * ay * v - az * s, ax * az * v + ay * s,
									  ay * ax * v + az * s, c + ay * ay * v,      ay * az * v - ax * s,
									  az * ax * v - ay * s, az * ay * v + ax * s, c + az * az * v};
				UNR for (int i = 0; i < 3; i++)
					UNR for (int j = 0; j < 3; j++) Tw[3 * i + j] = R0[3 * i] * Rq[j] + R0[3 * i + 1] * Rq[3 + j] + R0[3 * i + 2] * Rq[6 + j];
			}
			UNR for (int e = 0; e < 3; e++) Tw[9 + e] = p0v[e];
		} else {
			UNR for (int e = 0; e < 9; e++) Tw[e] = R0[e];
			const double a[3] = {ax, ay, az};
			double d[3];
			oct_mat3_vec(R0, a, d);
			UNR for (int e = 0; e < 3; e++) Tw[9 + e] = p0v[e] + d[e] * q_r;
		}
	}
	sm.vec[0][r] = dq_r;
	sm.vec[10][r] = gl0;
	sm.vec[11][r] = gl1;
	sm.vec[12][r] = gl2;
	sm.vec[5][r] = gl3;
	if (FULL && roleB) {
		lawc[lane] = lc0;
		if (lane < 26) lawc[64 + lane] = lc1;
	}
	const int mf_dec = __builtin_amdgcn_readfirstlane(mf_dec_v), jt_dec = __builtin_amdgcn_readfirstlane(jt_dec_v);
	OCT_REQUEST_STATE(_l)  // (dead code in the one-wavefront form)
	if (ROLE == 1) {
		UNR for (int e = 0; e < 3; e++) comv[e] = md.com[rr][e];
		UNR for (int e = 0; e < 6; e++) inev[e] = md.inertia[rr][e];
		mass_r = md.mass[rr];
	}
#undef OCT_REQUEST_STATE
	const bool track_mf = ROLE == 0 ? track_mf_e : track_mf_l, track_jt = ROLE == 0 ? track_jt_e : track_jt_l;
	const double mfi_pre = ROLE == 0 ? mfi_pre_e : mfi_pre_l, jti_pre = ROLE == 0 ? jti_pre_e : jti_pre_l;
	const double jkp_r = ROLE == 0 ? jkp_r_e : jkp_r_l, jkv_r = ROLE == 0 ? jkv_r_e : jkv_r_l, jki_r = ROLE == 0 ? jki_r_e : jki_r_l;
	OCT_FRAME_STEP(1, Tw, r)
	OCT_FRAME_STEP(2, Tw, r)
	OCT_FRAME_STEP(4, Tw, r)
	UNR for (int e = 0; e < 12; e++) sm.X[r][e] = Tw[e];  // the motion-force task reads its body's frame from here
	if (roleB && track_mf && r < 6) sm.ist[r] = mfi_pre;  // (behind the frames: the loads have had their time)
	if (roleA && GJ == 1 && track_jt && r < 4) sm.ist[6 + r] = jti_pre;
	// this lane's world frame is in Tw (padding lane 7: unused)
	const bool rev = jtype_r == 1;
	double z[3], o[3];
	{
		const double a[3] = {axv[0], axv[1], axv[2]};
		oct_mat3_vec(Tw, a, z);
		UNR for (int e = 0; e < 3; e++) o[e] = Tw[9 + e];
	}
	// per-body spatial inertia about the world origin
	double Ic[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	// this joint's motion vector about the world origin, S_r = (w, v): revolute (z, o x z), prismatic (0, z) -- what M_ir = S_i . (Ic_r S_r) needs
	// of the other joints (LDS stores are not guarded: the padding lane owns slot 7 of every row and nobody reads it)
	const double sw[3] = {rev ? z[0] : 0.0, rev ? z[1] : 0.0, rev ? z[2] : 0.0};
	const double sv[3] = {rev ? (o[1] * z[2] - o[2] * z[1]) : z[0], rev ? (o[2] * z[0] - o[0] * z[2]) : z[1], rev ? (o[0] * z[1] - o[1] * z[0]) : z[2]};
	UNR for (int e = 0; e < 3; e++) {
		sm.zo[r][e] = sw[e];
		sm.zo[r][3 + e] = sv[e];
	}
	if (roleA) {
		double c[3], Bd[10];
		const double cl[3] = {comv[0], comv[1], comv[2]};
		oct_mat3_vec(Tw, cl, c);
		UNR for (int e = 0; e < 3; e++) c[e] += o[e];
		const double* I6 = inev;
		const double Il[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
		double RI[9], Iw[9];
		UNR for (int i = 0; i < 3; i++)
			UNR for (int j = 0; j < 3; j++) RI[3 * i + j] = Tw[3 * i] * Il[j] + Tw[3 * i + 1] * Il[3 + j] + Tw[3 * i + 2] * Il[6 + j];
		UNR for (int i = 0; i < 3; i++)
			UNR for (int j = 0; j < 3; j++) Iw[3 * i + j] = RI[3 * i] * Tw[3 * j] + RI[3 * i + 1] * Tw[3 * j + 1] + RI[3 * i + 2] * Tw[3 * j + 2];
		const double m = mass_r, cc2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
		Bd[0] = m;
		Bd[1] = m * c[0];
		Bd[2] = m * c[1];
		Bd[3] = m * c[2];
		Bd[4] = Iw[0] + m * (cc2 - c[0] * c[0]);
		Bd[5] = Iw[4] + m * (cc2 - c[1] * c[1]);
		Bd[6] = Iw[8] + m * (cc2 - c[2] * c[2]);
		Bd[7] = Iw[1] - m * c[0] * c[1];
		Bd[8] = Iw[2] - m * c[0] * c[2];
		Bd[9] = Iw[5] - m * c[1] * c[2];
		UNR for (int e = 0; e < 10; e++) Ic[e] = act ? Bd[e] : 0.0;
	}
	// composite inertia of the subtree of joint r (serial chain: bodies r..6): suffix sums over the instance's lanes by DPP row_shl
	if (roleA) {
		// (selects, not multiplications by 0: a neighbouring instance with non-finite state must not leak in)
		UNR for (int e = 0; e < 10; e++) {
			const double up = OCT_ILV ? octl_shl<1>(Ic[e]) : oct_dpp<DPP_ROW_SHL + 1>(Ic[e]);
			Ic[e] += (OCT_ILV || r + 1 < 8) ? up : 0.0;
		}
		UNR for (int e = 0; e < 10; e++) {
			const double up = OCT_ILV ? octl_shl<2>(Ic[e]) : oct_dpp<DPP_ROW_SHL + 2>(Ic[e]);
			Ic[e] += (OCT_ILV || r + 2 < 8) ? up : 0.0;
		}
		UNR for (int e = 0; e < 10; e++) {
			const double up = OCT_ILV ? octl_shl<4>(Ic[e]) : oct_dpp<DPP_ROW_SHL + 4>(Ic[e]);
			Ic[e] += (OCT_ILV || r + 4 < 8) ? up : 0.0;
		}
	}
	oct_sync<DUO>();
	STAMP(2);
	// ---------------------------------------------------------------- M(q): column rr, M_i,rr = S_i . (Ic_rr S_rr), i <= rr
	if (roleA) {
		const double* w = sw;
		const double* v = sv;
		const double m = Ic[0], hx = Ic[1], hy = Ic[2], hz = Ic[3];
		const double f[3] = {m * v[0] + (w[1] * hz - w[2] * hy), m * v[1] + (w[2] * hx - w[0] * hz), m * v[2] + (w[0] * hy - w[1] * hx)};
		const double nO[3] = {Ic[4] * w[0] + Ic[7] * w[1] + Ic[8] * w[2] + (hy * v[2] - hz * v[1]),
							  Ic[7] * w[0] + Ic[5] * w[1] + Ic[9] * w[2] + (hz * v[0] - hx * v[2]),
							  Ic[8] * w[0] + Ic[9] * w[1] + Ic[6] * w[2] + (hx * v[1] - hy * v[0])};
		// row r of the lower triangle, M[r][i] for i <= r: M_ir = S_i . (Ic_r S_r) with the motion vectors S_i = (w_i, v_i) the lanes left in
		// LDS (one form for revolute and prismatic joints); entries beyond the diagonal are written too but never read
		double mrow[N];
		UNR for (int i = 0; i < N; i++)
			mrow[i] = sm.zo[i][0] * nO[0] + sm.zo[i][1] * nO[1] + sm.zo[i][2] * nO[2] + sm.zo[i][3] * f[0] + sm.zo[i][4] * f[1] + sm.zo[i][5] * f[2];
		UNR for (int i = 0; i < N; i++) sm.M[r][i] = mrow[i];
	}
	// ---------------------------------------------------------------- MotionForceTask: Jacobian column of this lane's joint
	// (two-wavefront form: B alone forms it, A takes the Jacobian from B's block behind the barrier below -- B gets there first)
	double pw[3] = {0, 0, 0}, Rc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, jwc[6] = {0, 0, 0, 0, 0, 0};  // control point, control frame, this joint's column of the world Jacobian [Jv; Jw]
	if (ROLE != 1) {
		const int bd = mf.body;
		double Rb[12];
		UNR for (int e = 0; e < 12; e++) Rb[e] = sm.X[bd][e];
		oct_mat3_vec(Rb, mf.pos, pw);
		UNR for (int e = 0; e < 3; e++) pw[e] += Rb[9 + e];
		UNR for (int i = 0; i < 3; i++)
			UNR for (int j = 0; j < 3; j++) Rc[3 * i + j] = Rb[3 * i] * mf.rot[j] + Rb[3 * i + 1] * mf.rot[3 + j] + Rb[3 * i + 2] * mf.rot[6 + j];
		const bool on = rr <= bd;
		const double rx = pw[0] - o[0], ry = pw[1] - o[1], rz = pw[2] - o[2];
		const double c0 = z[1] * rz - z[2] * ry, c1 = z[2] * rx - z[0] * rz, c2 = z[0] * ry - z[1] * rx;
		const double jw[6] = {on ? (rev ? c0 : z[0]) : 0.0, on ? (rev ? c1 : z[1]) : 0.0, on ? (rev ? c2 : z[2]) : 0.0,
							  (on && rev) ? z[0] : 0.0,    (on && rev) ? z[1] : 0.0,    (on && rev) ? z[2] : 0.0};
		UNR for (int a = 0; a < 6; a++) jwc[a] = act ? jw[a] : 0.0;
		if (FULL && !mf.bm_identity) {  // partial task: rows of the Jacobian in the basis of range(P), Jb = Bm^T Jw (columns >= k of Bm are zero)
			UNR for (int a = 0; a < 6; a++) {
				double sj = 0.0;
				UNR for (int e = 0; e < 6; e++) sj = fma(lawc[e * 6 + a], jw[e], sj);
				sm.J[a][r] = sj;
			}
		} else {
			UNR for (int a = 0; a < 6; a++) sm.J[a][r] = jw[a];
		}
	}
	// control law of this lane's joint (JointTask.cpp:299-345; full joint task, GJ != 1): it depends on the inputs only, so it runs here, behind the
	// mass matrix (its operands -- requested with the inputs -- have long arrived), and not in the tail of the cycle where its gain loads -- per-lane loads that cannot be hoisted above
	// the fences in between -- would sit on the critical path
	double fi_pre = 0.0;
	if (roleA && GJ != 1) {
		const double e = q_r - jg_q;
		double ie = track_jt ? jti_pre : 0.0;
		ie += e * jt.dt;  // :323-324
		jt_ie_new = ie;
		if (jt.vel_sat) {  // :327-341
			double vdes = -jkp_r * jt.kvinv[rr] * e - jki_r * jt.kvinv[rr] * ie;
			vdes = fmin(fmax(vdes, -jt.sat[rr]), jt.sat[rr]);
			fi_pre = -jkv_r * (dq_r - vdes);
		} else {
			fi_pre = -jkp_r * e - jkv_r * (dq_r - jg_dq) - jki_r * ie;  // :342-345
		}
	}
#if !defined(SAIP_OCT_NO_BROWS)
	if (BROWS) {
		if (ROLE == 2) flag_set(0, 1);  // B: the Jacobian is in its block
		oct_sync<DUO>();
	} else
#endif
	if (DUO && !trunc_mode) __syncthreads();  // workgroup barrier: B's Jacobian is complete (A arrives ~1 k clocks after B)
	else oct_sync<DUO>();                     // (reduced tasks: A waits further down, for the Jacobian B has reduced)
	STAMP(3);
	// ---------------------------------------------------------------- M^-1: Cholesky factor in every lane, own column of the inverse
	// M_BIE = M + sum_e c_e e e^T, the diagonal clamped from below (SingularityHandler.cpp:176-182).  Usual case (at most one clamped
	// entry per instance, e.g. the last link of an arm): nothing is factored twice -- with m = column e of M^-1 and
	// beta = c / (1 + c m_e), M_BIE^-1 = M^-1 - beta m m^T, hence J M_BIE^-1 J^T = A - beta t t^T (t = J m) and, by Sherman-Morrison,
	// Lambda_mod = Lambda + gamma (Lambda t)(Lambda t)^T with gamma = beta / (1 - beta t^T Lambda t).  Several clamped entries in some
	// instance of the wavefront (wave-uniform): M_BIE is factored alongside M and everything is formed twice.
	double mi[N] = {}, mb[N] = {};
	int eclamp = 0;
	double cclamp = 0.0;
	bool general_bie = false;
	if (roleA) {
		double L[N][N], dinv[N];
		UNR for (int i = 0; i < N; i++)
			UNR for (int j = 0; j <= i; j++) L[i][j] = sm.M[i][j];
		if (P.any_bie) {
			int nclamp = 0;
			UNR for (int e = 0; e < N; e++) {
				const double ce = P.bie_thr - L[e][e];
				if (ce > 0.0) {
					nclamp++;
					eclamp = e;
					cclamp = ce;
				}
			}
			general_bie = __any(nclamp > 1);
		}
		if (general_bie) {
			double Lb[N][N], dinvb[N];
			UNR for (int i = 0; i < N; i++)
				UNR for (int j = 0; j <= i; j++) Lb[i][j] = (i == j) ? fmax(L[i][j], P.bie_thr) : L[i][j];
			oct_cholesky<N>(L, dinv);
			oct_cholesky<N>(Lb, dinvb);
			oct_solve_unit<N>(L, dinv, rr, mi);
			oct_solve_unit<N>(Lb, dinvb, rr, mb);
		} else {
			oct_cholesky<N>(L, dinv);
			oct_solve_unit<N>(L, dinv, rr, mi);
			UNR for (int j = 0; j < N; j++) mb[j] = mi[j];
		}
	}
	if (ROLE == 1) {
		UNR for (int j = 0; j < N; j++) sm.N1[r][j] = mi[j];
	}
	const bool rank1_bie = P.any_bie && !general_bie;
	if (roleA && rank1_bie && r == eclamp) {
		UNR for (int j = 0; j < N; j++) sm.vec[3][j] = mi[j];  // column e of M^-1 for the whole instance (read after the next barrier)
	}
	STAMP(4);
	double Jf[6][N];
	unsigned keep_bits_b = 0x3fu;
	if (DUO && trunc_mode && ROLE == 1) {  // the Jacobian B has reduced, and which directions it kept (B is there ~15 k clocks later: its eigen-solve)
		__syncthreads();
		keep_bits_b = (unsigned)smB.vec[6][7];
	}
#if !defined(SAIP_OCT_NO_BROWS)
	if (BROWS && ROLE == 1) lost = flag_wait(0) == 0;
#endif
	UNR for (int a = 0; a < 6; a++)
		UNR for (int j = 0; j < N; j++) Jf[a][j] = (ROLE == 1 ? smB : sm).J[a][j];
	double vw[6] = {0, 0, 0, 0, 0, 0};
	if (roleB) {
		double dqa[N];
		UNR for (int j = 0; j < N; j++) dqa[j] = sm.vec[0][j];
		UNR for (int a = 0; a < 6; a++) {
			double s = 0.0;
			UNR for (int j = 0; j < N; j++) s = fma(Jf[a][j], dqa[j], s);
			vw[a] = s;
		}
		if (FULL && !mf.bm_identity) {  // the task velocities use the UNprojected Jacobian (MotionForceTask.cpp:293-298): vw = Jw dq
			UNR for (int a = 0; a < 6; a++) vw[a] = octl_sum(jwc[a] * dq_r);
		}
	}
	STAMP(5);
	STAMP(6);
	// ---- SingularityHandler branch predicate on G = J J^T (N_prec = I for the first task), every lane of the instance alike
	bool singular = false, truncated = false, blended_i = false;
#if defined(SAIP_OCT_EXP_NO_BLEND)  // diagnostic build only: what the blended block costs the ordinary path
	const bool strategies_on = false;
#else
	const bool strategies_on = GJ == 0 && mf.sing_handling && mf.sing_strategies && mf.sh != nullptr;  // batch-uniform; the blended branch lives in the headline stack's tail
#endif
	double G[6][6], U6[6][6];
	bool keepm[6] = {true, true, true, true, true, true};
	if (roleB) {
		double tr = 0.0;
		UNR for (int a = 0; a < 6; a++)
			UNR for (int c = 0; c <= a; c++) {
				double s = 0.0;
				UNR for (int j = 0; j < N; j++) s = fma(Jf[a][j], Jf[c][j], s);
				G[a][c] = s;
				G[c][a] = s;
				if (a == c) tr += s;
			}
		const double itr = oct_rcp(tr);
		const double smax2 = mf.s_max * mf.s_max, tol2 = mf.s_abs_tol * mf.s_abs_tol;
		auto ldl_positive = [&](const double shift) {
			double Gs[6][6];
			UNR for (int a = 0; a < 6; a++)
				UNR for (int c = 0; c <= a; c++) Gs[a][c] = G[a][c] * itr;
			bool pos = true;
			UNR for (int a = 0; a < 6; a++) Gs[a][a] = (a < kmf) ? Gs[a][a] - shift : 1.0;  // zero rows of a partial task: padded
			UNR for (int kk = 0; kk < 6; kk++) {
				const double d = Gs[kk][kk];
				pos = pos && (d > 1e-13);
				const double id = oct_rcp(d);
				UNR for (int i = kk + 1; i < 6; i++) {
					const double lik = Gs[i][kk] * id;
					UNR for (int j = kk + 1; j <= i; j++) Gs[i][j] = fma(-lik, Gs[j][kk], Gs[i][j]);
				}
			}
			return pos;
		};
		// stage 1: lambda_max(Gs) <= trace(Gs) = 1, lambda_max(G) >= trace(G)/6; stage 2: trace-power bound (see saip_kernel_lane.hip)
		bool need_exact = !((tr > 0.0) && (tr >= (double)kmf * tol2) && ldl_positive(smax2));
#if defined(SAIP_OCT_FORCE_EXACT)  // diagnostic build only: every instance takes the exact fallbacks (validated against the oracle once per change)
		need_exact = true;
#endif
		if (__any(need_exact)) {
			double G2[6][6], Gt[6][6];
			UNR for (int a = 0; a < 6; a++)
				UNR for (int c = 0; c < 6; c++) G2[a][c] = G[a][c] * itr;
			UNR for (int rep = 0; rep < 2; rep++) {
				UNR for (int a = 0; a < 6; a++)
					UNR for (int c = 0; c <= a; c++) {
						double s = 0.0;
						UNR for (int l = 0; l < 6; l++) s = fma(G2[a][l], G2[l][c], s);
						Gt[a][c] = s;
					}
				UNR for (int a = 0; a < 6; a++)
					UNR for (int c = 0; c <= a; c++) {
						G2[a][c] = Gt[a][c];
						G2[c][a] = Gt[a][c];
					}
			}
			double t8 = 0.0;
			UNR for (int a = 0; a < 6; a++)
				UNR for (int c = 0; c < 6; c++) t8 = fma(G2[a][c], G2[a][c], t8);
			const double u = sqrt(sqrt(sqrt(t8)));
			const bool ok_s0 = (u * mf.cert_kroot * tr >= tol2);
			const bool pos = ldl_positive(smax2 * u);
			need_exact = need_exact && !(ok_s0 && pos && tr > 0.0);
			if (need_exact) {
				// stage 3, the opposite certificate: with l = max_i Gs_ii <= lambda_max(Gs), a NEGATIVE LDL^T pivot of Gs - s_max^2 l I (all earlier
				// pivots positive: Sylvester) proves lambda_min < s_max^2 lambda_max, i.e. the instance is outside the non-singular branch for
				// sure; so does trace(G) < s_abs_tol^2.  Batches with many near-singular instances would otherwise send every one of them
				// through the eigen-solve only to confirm it (measured: 70 us instead of 10 us per launch at a third of the batch).
				double lmx = 0.0;
				UNR for (int a = 0; a < 6; a++) lmx = (a < kmf) ? fmax(lmx, G[a][a] * itr) : lmx;
				double Gs[6][6];
				UNR for (int a = 0; a < 6; a++)
					UNR for (int c = 0; c <= a; c++) Gs[a][c] = G[a][c] * itr;
				UNR for (int a = 0; a < 6; a++) Gs[a][a] = (a < kmf) ? Gs[a][a] - smax2 * lmx : 1.0;
				bool neg = false, clean = true;
				UNR for (int kk = 0; kk < 6; kk++) {
					const double dpv = Gs[kk][kk];
					neg = neg || (clean && dpv < -1e-12);
					clean = clean && (dpv > 1e-13);
					const double id = oct_rcp(dpv);
					UNR for (int i = kk + 1; i < 6; i++) {
						const double lik = Gs[i][kk] * id;
						UNR for (int j = kk + 1; j <= i; j++) Gs[i][j] = fma(-lik, Gs[j][kk], Gs[i][j]);
					}
				}
				if ((tr > 0.0 && neg) || tr < tol2) {
					singular = true;
					need_exact = false;
				}
			}
#if defined(SAIP_OCT_FORCE_EXACT)
			need_exact = true;
#endif
			if (strategies_on && need_exact) {
				// undecided by the certificates: the blended branch below runs the eigen-solve anyway and hands the instance back to the
				// ordinary result when all six directions turn out to be kept ("clear")
				singular = true;
				need_exact = false;
			}
			if (__any(need_exact)) {  // exact path (rare): singular values from the Gram eigenvalues, one lane per instance (SingularityHandler.cpp:78-121)
				if (need_exact && r == 0) {
					for (int a = 0; a < 6; a++)
						for (int c = 0; c < 6; c++) sm.A[a][c] = G[a][c];
					oct_serial_jacobi(&sm.A[0][0], nullptr, kmf, 8);
					double lmax = 0.0;
					for (int i = 0; i < kmf; i++) lmax = fmax(lmax, sm.A[i][i]);
					const double s0 = sqrt(lmax);
					bool sing = !(s0 >= mf.s_abs_tol);
					for (int i = 0; i < kmf; i++)
						if (sqrt(fmax(sm.A[i][i], 0.0)) / s0 < mf.s_max) sing = true;
					sm.vec[9][0] = sing ? 1.0 : 0.0;
				}
				oct_sync<DUO>();
				if (need_exact) singular = sm.vec[9][0] != 0.0;
				oct_sync<DUO>();
			}
		}
	}
	if (trunc_mode) {
		UNR for (int i = 0; i < 6; i++)
			UNR for (int j = 0; j < 6; j++) U6[i][j] = (i == j) ? 1.0 : 0.0;
	}
	if (trunc_mode && __any(singular)) {
		// disableSingularityHandling(): _N = _N_ns and tau = tau_ns (SingularityHandler.cpp:146-148, 319-330): the task acts on the left singular
		// directions with sigma_i / sigma_0 >= s_max (the count of those = the first index below s_max of the sorted spectrum, :100-121).
		// Eigen-pairs of G = J J^T by the register Jacobi solve, in every lane of the wavefront (it only runs when some instance needs it).
		double E[6][6], Ue[6][6];
		UNR for (int i = 0; i < 6; i++)
			UNR for (int j = 0; j < 6; j++) E[i][j] = G[i][j];
		oct_jacobi_n<6>(E, Ue, lane);
		double lmax = 0.0;
		UNR for (int c = 0; c < 6; c++) lmax = fmax(lmax, E[c][c]);
		const double s0 = sqrt(lmax);
		int ns = 0;
		bool kp[6];
		UNR for (int c = 0; c < 6; c++) {
			kp[c] = (s0 >= mf.s_abs_tol) && (sqrt(fmax(E[c][c], 0.0)) / s0 >= mf.s_max);
			ns += kp[c] ? 1 : 0;
		}
		const bool reduce = singular && ns >= 2 && ns < 6;  // fewer than two directions left (or a fully singular task): stays flagged, the general kernel behind handles it
		if (reduce) {
			singular = false;
			truncated = true;
			UNR for (int c = 0; c < 6; c++) keepm[c] = kp[c];
			UNR for (int i = 0; i < 6; i++)
				UNR for (int j = 0; j < 6; j++) U6[i][j] = Ue[i][j];
		}
		// rows of the reduced Jacobian: J_ns = U_ns^T J, dropped directions zero (their diagonals are padded below)
		double jc6[6], jn[6];
		UNR for (int i = 0; i < 6; i++) jc6[i] = sm.J[i][rr];
		UNR for (int a = 0; a < 6; a++) {
			double sj = 0.0;
			UNR for (int i = 0; i < 6; i++) sj = fma(U6[i][a], jc6[i], sj);
			jn[a] = keepm[a] ? sj : 0.0;
		}
		oct_sync<DUO>();
		UNR for (int a = 0; a < 6; a++) sm.J[a][r] = jn[a];
		oct_sync<DUO>();
		UNR for (int a = 0; a < 6; a++)
			UNR for (int j = 0; j < N; j++) Jf[a][j] = sm.J[a][j];
	}
	if (DUO && trunc_mode) {
		if (ROLE == 2) {
			unsigned kb = 0;
			UNR for (int c = 0; c < 6; c++) kb |= keepm[c] ? (1u << c) : 0u;
			if (r == 0) sm.vec[6][7] = (double)kb;
			__syncthreads();  // (unconditional: every wavefront B arrives, whether or not it had anything to reduce)
		} else if (ROLE == 1) {
			UNR for (int c = 0; c < 6; c++) keepm[c] = ((keep_bits_b >> c) & 1u) != 0u;
		}
	}
	bool keep_ra = true;  // is this lane's row of the task kept (partial task: rows >= k are not; reduced task: the dropped directions)
	STAMP(7);
	// ---- operationalSpaceMatrices(J): this lane's COLUMN of T1 = J M^-1 (M^-1 symmetric: its column rr is the row this lane solved for),
	// then its ROW of A = T1 J^T; the same with M_BIE^-1
	double mcol[N], beta = 0.0;
	UNR for (int j = 0; j < N; j++) mcol[j] = 0.0;
	const int ra = r < 6 ? r : 5;
	UNR for (int a = 0; a < 6; a++) keep_ra = (a == ra) ? (keepm[a] && a < kmf) : keep_ra;
	double t1c[6] = {0, 0, 0, 0, 0, 0};
	if (roleA) {
		double tbc[6];
		UNR for (int a = 0; a < 6; a++) {
			double s1 = 0.0, s2 = 0.0;
			UNR for (int l = 0; l < N; l++) {
				s1 = fma(Jf[a][l], mi[l], s1);
				if (general_bie) s2 = fma(Jf[a][l], mb[l], s2);
			}
			t1c[a] = s1;
			tbc[a] = s2;
		}
		UNR for (int a = 0; a < 6; a++) {
			sm.T1[a][r] = t1c[a];
			if (general_bie) sm.Lam[a][r] = tbc[a];
		}
#if !defined(SAIP_OCT_NO_BROWS)
		const bool rows_by_b = BROWS && ROLE == 1 && !general_bie;
		if (BROWS && ROLE == 1) flag_set(1, general_bie ? 2 : 1);
#else
		const bool rows_by_b = false;
#endif
		oct_sync<DUO>();
		if (rank1_bie) {
			double mee = 0.0;
			UNR for (int j = 0; j < N; j++) {
				mcol[j] = sm.vec[3][j];
				mee = (j == eclamp) ? mcol[j] : mee;
			}
			beta = cclamp * oct_rcp(fma(cclamp, mee, 1.0));
		}
		double t1[N], tb[N];
		UNR for (int j = 0; j < N; j++) {
			t1[j] = sm.T1[ra][j];
			tb[j] = general_bie ? sm.Lam[ra][j] : 0.0;
		}
		if (!rows_by_b) {
		UNR for (int c = 0; c < 6; c++) {
			double s = 0.0, s2 = 0.0;
			UNR for (int j = 0; j < N; j++) {
				s = fma(t1[j], Jf[c][j], s);
				if (general_bie) s2 = fma(tb[j], Jf[c][j], s2);
			}
			const bool padrow = FULL && !keep_ra && c == ra;  // zero row of a partial / reduced task: keep the factorisation regular
			sm.A[r][c] = padrow ? 1.0 : s;
			if (general_bie) sm.Am[r][c] = padrow ? 1.0 : s2;
		}
		}
		if (ROLE == 1) {  // what the blended block (run by B) wants from this wavefront besides A, T1, M and the columns of M^-1
			if (general_bie) {
				UNR for (int j = 0; j < N; j++) sm.Lam[r][j] = mb[j];  // (its rows were consumed above)
			}
			if (r == 0) {
				sm.vec[9][0] = beta;
				sm.vec[9][1] = (double)eclamp;
				sm.vec[9][2] = general_bie ? 1.0 : 0.0;
			}
		}
		oct_sync<DUO>();
	}
	STAMP(8);
	double lam[6], lmod[6];
	// headline stack (GJ == 0): nobody needs Lambda itself, only its action on three vectors -- every lane keeps the Cholesky factor of
	// A = J M^-1 J^T and solves (no row of the inverse, no exchange of Lambda between the lanes)
	double L6[6][6], dinv6[6];
	if (GJ == 0) {
		if (!DUO) {
			UNR for (int i = 0; i < 6; i++)
				UNR for (int j = 0; j <= i; j++) L6[i][j] = sm.A[i][j];
			oct_cholesky<6>(L6, dinv6);
		}
	} else {
		double L[6][6], dinv[6];
		UNR for (int i = 0; i < 6; i++)
			UNR for (int j = 0; j <= i; j++) L[i][j] = sm.A[i][j];
		oct_cholesky<6>(L, dinv);
		oct_solve_unit<6>(L, dinv, ra, lam);
		if (mf_dec == DEC_FULL) {
			UNR for (int c = 0; c < 6; c++) lmod[c] = lam[c];
		} else if (mf_dec == DEC_IMPEDANCE) {
			UNR for (int c = 0; c < 6; c++) lmod[c] = (c == ra) ? 1.0 : 0.0;
		} else if (general_bie) {
			UNR for (int i = 0; i < 6; i++)
				UNR for (int j = 0; j <= i; j++) L[i][j] = sm.Am[i][j];
			oct_cholesky<6>(L, dinv);
			oct_solve_unit<6>(L, dinv, ra, lmod);
		} else {  // rank-one form (beta = 0 when nothing is clamped: Lambda_mod = Lambda)
			double t[6], lt_own = 0.0;
			UNR for (int a = 0; a < 6; a++) {
				double sacc = 0.0;
				UNR for (int l = 0; l < N; l++) sacc = fma(Jf[a][l], mcol[l], sacc);
				t[a] = sacc;
				lt_own = fma(lam[a], sacc, lt_own);  // (Lambda t)_ra: Lambda symmetric, lam = its row ra
			}
			sm.vec[4][r] = lt_own;
			oct_sync<DUO>();
			double lt[6], q = 0.0;
			UNR for (int a = 0; a < 6; a++) {
				lt[a] = sm.vec[4][a];
				q = fma(t[a], lt[a], q);
			}
			const double gamma = beta * oct_rcp(fma(-beta, q, 1.0));
			UNR for (int c = 0; c < 6; c++) lmod[c] = fma(gamma * lt_own, lt[c], lam[c]);
		}
	}
	if (GJ != 0) {
		UNR for (int c = 0; c < 6; c++) sm.Lam[r][c] = lam[c];
	}
	STAMP(9);
	double Fum6[6], Ff6[6] = {0, 0, 0, 0, 0, 0};  // GJ == 0: the task forces, identical in every lane of the instance
	// ---- control law (MotionForceTask.cpp:286-509 with sigma_force = sigma_moment = 0, OTG off), every lane alike
	if (roleB) {
		double ip[3] = {0, 0, 0}, io[3] = {0, 0, 0}, Fum[6], G24[24];
		UNR for (int e = 0; e < 24; e++) G24[e] = sm.vec[10 + (e >> 3)][e & 7];
		if (track_mf) {
			UNR for (int i = 0; i < 3; i++) {
				ip[i] = sm.ist[i];
				io[i] = sm.ist[3 + i];
			}
		}
		if (FULL && mf.general_law) {
			// velocity saturation and / or force & moment spaces (open-loop force control): the shared general law, MotionForceTask.cpp:306-487
			double G30[30], Ff[6] = {0, 0, 0, 0, 0, 0};
			UNR for (int e = 0; e < 24; e++) G30[e] = G24[e];
			UNR for (int e = 0; e < 6; e++) G30[24 + e] = sm.vec[5][e];
			law_motion_force_general<false>(&mf, vw, pw, Rc, G30, ip, io, (double*)nullptr, Fum, Ff);
			UNR for (int a = 0; a < 6; a++) Ff6[a] = Ff[a];
		} else if (FULL && !mf.bm_identity) {
			// partial task: the short law with its projections (MotionForceTask.cpp:286-509 with sigma_force = sigma_moment = 0), then
			// F in the basis of range(P): fhat = Bm^T F_um
			double v[3], w[3], e3[3], tmp[3], Fw[6];
			UNR for (int i = 0; i < 3; i++) {
				v[i] = lawc[36 + 3 * i] * vw[0] + lawc[36 + 3 * i + 1] * vw[1] + lawc[36 + 3 * i + 2] * vw[2];
				w[i] = lawc[45 + 3 * i] * vw[3] + lawc[45 + 3 * i + 1] * vw[4] + lawc[45 + 3 * i + 2] * vw[5];
			}
			UNR for (int i = 0; i < 3; i++) e3[i] = pw[i] - G24[i];
			UNR for (int i = 0; i < 3; i++) ip[i] += (lawc[54 + 3 * i] * e3[0] + lawc[54 + 3 * i + 1] * e3[1] + lawc[54 + 3 * i + 2] * e3[2]) * mf.dt;
			UNR for (int i = 0; i < 3; i++) tmp[i] = G24[18 + i] - lawc[72 + i] * e3[i] - lawc[75 + i] * (v[i] - G24[12 + i]) - lawc[78 + i] * ip[i];
			UNR for (int i = 0; i < 3; i++) Fw[i] = lawc[54 + 3 * i] * tmp[0] + lawc[54 + 3 * i + 1] * tmp[1] + lawc[54 + 3 * i + 2] * tmp[2];
			double oe[3] = {0, 0, 0};
			UNR for (int c = 0; c < 3; c++) {
				const double a0 = Rc[c], a1 = Rc[3 + c], a2 = Rc[6 + c], b0 = G24[3 + c], b1 = G24[6 + c], b2 = G24[9 + c];
				oe[0] -= 0.5 * (a1 * b2 - a2 * b1);
				oe[1] -= 0.5 * (a2 * b0 - a0 * b2);
				oe[2] -= 0.5 * (a0 * b1 - a1 * b0);
			}
			double dphi[3];
			UNR for (int i = 0; i < 3; i++) dphi[i] = lawc[63 + 3 * i] * oe[0] + lawc[63 + 3 * i + 1] * oe[1] + lawc[63 + 3 * i + 2] * oe[2];
			UNR for (int i = 0; i < 3; i++) io[i] += dphi[i] * mf.dt;
			UNR for (int i = 0; i < 3; i++) tmp[i] = G24[21 + i] - lawc[81 + i] * dphi[i] - lawc[84 + i] * (w[i] - G24[15 + i]) - lawc[87 + i] * io[i];
			UNR for (int i = 0; i < 3; i++) Fw[3 + i] = lawc[63 + 3 * i] * tmp[0] + lawc[63 + 3 * i + 1] * tmp[1] + lawc[63 + 3 * i + 2] * tmp[2];
			UNR for (int a = 0; a < 6; a++) {
				double sf = 0.0;
				UNR for (int e = 0; e < 6; e++) sf = fma(lawc[0 + e * 6 + a], Fw[e], sf);
				Fum[a] = sf;
			}
		} else {
		// full 6-dof task without force space: the task projections and sigma_position / sigma_orientation are identities (oct_eligible)
		double e3[3];
		UNR for (int i = 0; i < 3; i++) e3[i] = pw[i] - G24[i];
		UNR for (int i = 0; i < 3; i++) ip[i] = fma(e3[i], mf.dt, ip[i]);
		UNR for (int i = 0; i < 3; i++) Fum[i] = G24[18 + i] - mf.kp_pos[i] * e3[i] - mf.kv_pos[i] * (vw[i] - G24[12 + i]) - mf.ki_pos[i] * ip[i];
		double oe[3] = {0, 0, 0};  // orientationError(desired, current) = -1/2 sum_c Rc[:,c] x Rd[:,c]
		UNR for (int c = 0; c < 3; c++) {
			const double a0 = Rc[c], a1 = Rc[3 + c], a2 = Rc[6 + c], b0 = G24[3 + c], b1 = G24[6 + c], b2 = G24[9 + c];
			oe[0] -= 0.5 * (a1 * b2 - a2 * b1);
			oe[1] -= 0.5 * (a2 * b0 - a0 * b2);
			oe[2] -= 0.5 * (a0 * b1 - a1 * b0);
		}
		UNR for (int i = 0; i < 3; i++) io[i] = fma(oe[i], mf.dt, io[i]);
		UNR for (int i = 0; i < 3; i++) Fum[3 + i] = G24[21 + i] - mf.kp_ori[i] * oe[i] - mf.kv_ori[i] * (vw[3 + i] - G24[15 + i]) - mf.ki_ori[i] * io[i];
		}
		if (track_mf && r == 0) {  // left in LDS: committed in the epilogue once the final status is known (the joint task may still flag the instance)
			UNR for (int i = 0; i < 3; i++) {
				sm.ist[i] = ip[i];
				sm.ist[3 + i] = io[i];
			}
		}
		if (trunc_mode) {  // task_range_ns^T F_um (SingularityHandler.cpp:321-322); the identity for instances that were not reduced
			double fh[6];
			UNR for (int a = 0; a < 6; a++) {
				double sf = 0.0;
				UNR for (int i = 0; i < 6; i++) sf = fma(U6[i][a], Fum[i], sf);
				fh[a] = keepm[a] ? sf : 0.0;
			}
			UNR for (int a = 0; a < 6; a++) Fum[a] = fh[a];
		}
		UNR for (int a = 0; a < 6; a++) Fum6[a] = Fum[a];
	}
	if (DUO) {
		// first workgroup barrier: B publishes the task force (and, for the stacks whose tail it does not share, the branch predicate); A has
		// left A = J M^-1 J^T, T1 and M^-1 in its block
		if (ROLE == 2) {
			if (r < 6) {
				double fo = 0.0, ffo = 0.0;
				UNR for (int a = 0; a < 6; a++) {
					fo = (a == r) ? Fum6[a] : fo;
					ffo = (a == r) ? Ff6[a] : ffo;
				}
				sm.vec[6][r] = fo;
				sm.vec[5][r] = ffo;
			}
			if (r == 6) sm.vec[6][6] = (singular ? 1.0 : 0.0) + (truncated ? 2.0 : 0.0);
		}
#if !defined(SAIP_OCT_NO_BROWS)
		if (BROWS && ROLE == 2) {
			const int f1 = flag_wait(1);
			lost = f1 == 0;
			if (f1 == 1) {  // the rows of A = T1 J^T, into A's block
				double t1[N];
				UNR for (int j = 0; j < N; j++) t1[j] = smA.T1[ra][j];
				UNR for (int c = 0; c < 6; c++) {
					double s = 0.0;
					UNR for (int j = 0; j < N; j++) s = fma(t1[j], Jf[c][j], s);
					smA.A[r][c] = s;
				}
			}
		}
#endif
		STAMP(29);
		__syncthreads();
		STAMP(30);
		if (ROLE == 2 && GJ != 0) return false;
		if (ROLE == 1) {
			UNR for (int a = 0; a < 6; a++) {
				Fum6[a] = smB.vec[6][a];
				Ff6[a] = smB.vec[5][a];
			}
			if (GJ != 0) {  // (headline stack: the final flags arrive at the second barrier)
				const int fl = (int)smB.vec[6][6];
				singular = (fl & 1) != 0;
				truncated = (fl & 2) != 0;
			}
		}
		if (GJ == 0) {
			// both factor A (B cannot wait for A's factor: the tail is split between them from here)
			UNR for (int i = 0; i < 6; i++)
				UNR for (int j = 0; j <= i; j++) L6[i][j] = smA.A[i][j];
			oct_cholesky<6>(L6, dinv6);
			if (ROLE == 2) {
				UNR for (int a = 0; a < 6; a++) t1c[a] = smA.T1[a][r];
				UNR for (int j = 0; j < N; j++) mi[j] = smA.N1[r][j];
			}
		}
	}
	if (GJ != 0) {
		// g_a = (Lambda_mod F_um)_a + (F_f)_a  (SingularityHandler.cpp:307-309); Bm = I for the full task
		double s = 0.0;
		UNR for (int a = 0; a < 6; a++) s = (a == ra) ? Ff6[a] : s;
		UNR for (int c = 0; c < 6; c++) s = fma(lmod[c], Fum6[c], s);
		sm.vec[2][r] = s;
		oct_sync<DUO>();
	}
	STAMP(10);
	// ---- tau_r = (J^T g)_r;  N_1 = I - M^-1 J^T Lambda J = I - T1^T Lambda J: this lane forms COLUMN rr (what the joint task's range
	// basis works on), N1[i][rr] = d_i,rr - sum_a T1[a][i] (Lambda J[:, rr])_a, and parks it as a row of the transposed matrix
	double tau_r;
	double cj[N];  // column rr of N_1
	double cn = 0.0;
	double tv;
	if (GJ == 0) {
		// ---- headline stack: full 6-dof task + full joint task.  g = Lambda_mod F_um + F_f by solves with the factor of A: FULL g = A^-1 F;
		// bounded inertia estimates with one clamped entry (A_mod = A - beta t t^T, t = J m): Sherman-Morrison on the solves,
		// g = y1 + gamma y2 (t . y1), y1 = A^-1 F, y2 = A^-1 t, gamma = beta / (1 - beta t . y2); several clamped entries: A_mod is
		// factored too.  tau_1 = J^T g in full in every lane (the joint task's disturbance compensation wants M^-1 tau_1).
		double g[6];
		double wr = 0.0;
		tau_r = 0.0;
		if (roleA) {
		if (mf_dec == DEC_IMPEDANCE) {
			UNR for (int a = 0; a < 6; a++) g[a] = Fum6[a] + Ff6[a];
		} else if (mf_dec == DEC_BIE && general_bie) {
			double Lb[6][6], dinvb[6], y[6];
			UNR for (int i = 0; i < 6; i++)
				UNR for (int j = 0; j <= i; j++) Lb[i][j] = sm.Am[i][j];
			oct_cholesky<6>(Lb, dinvb);
			oct_solve<6>(Lb, dinvb, Fum6, y);
			UNR for (int a = 0; a < 6; a++) g[a] = y[a] + Ff6[a];
		} else {
			double y1[6];
			oct_solve<6>(L6, dinv6, Fum6, y1);
			if (mf_dec == DEC_BIE) {  // rank-one form (beta = 0 when nothing is clamped)
				double t[6], y2[6], ty1 = 0.0, ty2 = 0.0;
				UNR for (int a = 0; a < 6; a++) {
					double sacc = 0.0;
					UNR for (int l = 0; l < N; l++) sacc = fma(Jf[a][l], mcol[l], sacc);
					t[a] = sacc;
				}
				oct_solve<6>(L6, dinv6, t, y2);
				UNR for (int a = 0; a < 6; a++) {
					ty1 = fma(t[a], y1[a], ty1);
					ty2 = fma(t[a], y2[a], ty2);
				}
				const double gamma = beta * oct_rcp(fma(-beta, ty2, 1.0));
				UNR for (int a = 0; a < 6; a++) g[a] = fma(gamma * ty1, y2[a], y1[a]) + Ff6[a];
			} else {
				UNR for (int a = 0; a < 6; a++) g[a] = y1[a] + Ff6[a];
			}
		}
		double tau1[N];
		UNR for (int j = 0; j < N; j++) {
			double s = 0.0;
			UNR for (int a = 0; a < 6; a++) s = fma(Jf[a][j], g[a], s);
			tau1[j] = s;
			wr = fma(mi[j], s, wr);  // (M^-1 tau_prec)_rr: M^-1 symmetric, mi = its column rr
		}
		UNR for (int j = 0; j < N; j++) tau_r = (j == rr) ? tau1[j] : tau_r;
		}  // roleA
		// ---- JointTask (S = I) in the nullspace of the 6-dof task, JointTask.cpp:218-356.  N_1 = I - M^-1 J^T Lambda J has rank one:
		// N_1 = z z^T M / (z^T M z), z spanning null(J).  Its ROW rr, e_rr - J^T (A^-1 t1c) with t1c = this lane's column of T1 = J M^-1,
		// needs nothing from the other lanes and is z_rr (M z)^T / (z^T M z).  The row of largest norm (largest |z_rr|) is broadcast:
		// v ~ M z, then z'_rr = (M^-1 v)_rr from this lane's column of M^-1.  With the range basis U = z' / |z'| (matrixRangeBasis of
		// Jp = N_1, one direction) the task reduces to scalars: Jh = U^T Jp = v^T / (|z'| s'), Lambda = (Jh M^-1 Jh^T)^-1 = z'.v / |z'|^2, and
		//   tau_joint = Jh^T (Lambda a1 + Lambda_mod b1) = v / |z'|^2 (z'.a + rho z'.f),   a = ddq_d - M^-1 tau_prec, f = the PD(I) force,
		// rho = Lambda_mod / Lambda: 1 (full decoupling), |z'|^2 / z'.v (impedance), z'.v / (z'.v - beta z'_e^2) (bounded inertia, one
		// clamped entry e: M_BIE^-1 = M^-1 - beta m m^T and m^T M z' = z'_e), z'.v / (v^T M_BIE^-1 v) (several clamped entries).
		double v[N] = {}, zr = 0.0, v_own = 0.0, zz = 0.0, zv = 0.0;
		if (roleB) {
		double x6[6], rowv[N], nr = 0.0;
		oct_solve<6>(L6, dinv6, t1c, x6);
		UNR for (int j = 0; j < N; j++) {
			double acc = (j == rr) ? 1.0 : 0.0;
			UNR for (int a = 0; a < 6; a++) acc = fma(-Jf[a][j], x6[a], acc);
			rowv[j] = acc;
			nr = fma(acc, acc, nr);
		}
		double bestw = act ? nr : -1.0;
		int jbw = r;
#define OCT_ARGMAX_STEP(CTRL)                                                                                   \
	{                                                                                                           \
		const double ob = oct_dpp<CTRL>(bestw);                                                                 \
		const int oj = __builtin_amdgcn_update_dpp(0, jbw, CTRL, 0xF, 0xF, true);                               \
		const bool take = ob > bestw || (ob == bestw && oj < jbw);                                              \
		bestw = take ? ob : bestw;                                                                              \
		jbw = take ? oj : jbw;                                                                                  \
	}
		OCT_ARGMAX_STEP(OCT_BFLY0)
		OCT_ARGMAX_STEP(OCT_BFLY1)
		OCT_ARGMAX_STEP(OCT_BFLY2)
#undef OCT_ARGMAX_STEP
		const int src = octl_src(lane, jbw);
		UNR for (int j = 0; j < N; j++) {
			v[j] = __shfl(rowv[j], src);
			zr = fma(mi[j], v[j], zr);
			v_own = (j == rr) ? v[j] : v_own;
		}
		zr = act ? zr : 0.0;
		zz = octl_sum(zr * zr);
		zv = octl_sum(zr * v_own);
		}  // roleB
		double fi = 0.0;
		if (roleA) {
		fi = fi_pre;  // (the joint control law ran behind M(q), see fi_pre)
		}
		double tvo = 0.0;
		bool ovr = false;
		if (ROLE == 0) {
		const double ai = jg_ddq - wr;
		const double za = octl_sum(zr * ai);
		const double zf = octl_sum(zr * fi);
		double rho = 1.0;
		if (jt_dec == DEC_IMPEDANCE) {
			rho = zz * oct_rcp(zv);
		} else if (jt_dec == DEC_BIE) {
			if (general_bie) {
				if (DUO) {
					UNR for (int j = 0; j < N; j++) v[j] = smB.vec[4][j];
				}
				double mbv = 0.0;
				UNR for (int j = 0; j < N; j++) mbv = fma(mb[j], v[j], mbv);  // (M_BIE^-1 v)_rr
				const double qb = octl_sum(act ? v_own * mbv : 0.0);
				rho = zv * oct_rcp(qb);
			} else {
				const double ze = octl_sum((act && r == eclamp) ? zr : 0.0);
				rho = zv * oct_rcp(fma(-beta * ze, ze, zv));
			}
		}
		tv = fma(v_own * oct_rcp(zz), fma(rho, zf, za), tau_r);  // tau += Jh^T g
		}
		// ------------------------------------------------------------------------------------------------------------------------------
		// Blended singularity strategies, in this kernel (SingularityHandler.cpp:100-121, 146-228, 230-295, 310-367; the reference's default
		// whenever the task is inside its singularity bounds).  Only wavefronts that hold such an instance come here.  Everything happens
		// in the eigenbasis of G = J J^T (register Jacobi solve; no sorting needed: the non-singular set K = { c : sigma_c >= s_max
		// sigma_max } is a prefix of the sorted spectrum, and everything else is invariant to the order inside K and inside S):
		//   J_u = U^T J,  A_u = U^T (J M^-1 J^T) U,  B = blockdiag(A_u[K,K], A_u[S,S]) -> one Cholesky gives Lambda_ns and Lambda_s,
		//   posture Jacobian V_s^T N_ns = D^-1 (J_S - A_SK Lambda_ns J_K), its op-space inertia = D^-1 (Schur complement of A_u) D^-1,
		//   N = N_js N_ns projects onto null(J) again, so the joint task behind keeps its closed rank-one form with z from U.
		if (ROLE != 1 && __builtin_expect(strategies_on && __any(singular), 0)) {
			if (ROLE == 2) {
				// wavefront B runs the block in the two-wavefront form (it is off A's critical path and A's registers stay out of it): what the
				// ordinary path left in A's block, and the joint task's goal and control law once more (no shadow integrator store here: A's)
				eclamp = (int)smA.vec[9][1];
				general_bie = smA.vec[9][2] != 0.0;
				beta = smA.vec[9][0];
				const bool r1b = P.any_bie && !general_bie;
				UNR for (int j = 0; j < N; j++) mcol[j] = r1b ? smA.vec[3][j] : 0.0;
				UNR for (int j = 0; j < N; j++) mb[j] = general_bie ? smA.Lam[r][j] : mi[j];
				jg_q = saip_ldg(P.goal2[1], (size_t)rr * ld + b);
				jg_dq = saip_ldg(P.goal2[1], (size_t)(N + rr) * ld + b);
				jg_ddq = saip_ldg(P.goal2[1], (size_t)(2 * N + rr) * ld + b);
				const double e = q_r - jg_q;
				double ie = 0.0;
				if (jt.has_ki || P.integ_always) ie = saip_ldg(P.integ2[1], (size_t)rr * ld + b);
				ie += e * jt.dt;
				if (jt.vel_sat) {
					double vdes = -jt.kp[rr] * jt.kvinv[rr] * e - jt.ki[rr] * jt.kvinv[rr] * ie;
					vdes = fmin(fmax(vdes, -jt.sat[rr]), jt.sat[rr]);
					fi = -jt.kv[rr] * (dq_r - vdes);
				} else {
					fi = -jt.kp[rr] * e - jt.kv[rr] * (dq_r - jg_dq) - jt.ki[rr] * ie;
				}
			}
			STAMP(16);
			double Ue[6][6], sig[6], ju[6];
			bool keep[6];
			double smax = 0.0;
			double trg = 0.0;
			{
				// (G = J J^T is formed again here rather than kept alive from the branch predicate: 72 registers less across the ordinary path)
				double E[6][6];
				UNR for (int a = 0; a < 6; a++)
					UNR for (int c = 0; c <= a; c++) {
						double sacc = 0.0;
						UNR for (int j = 0; j < N; j++) sacc = fma(Jf[a][j], Jf[c][j], sacc);
						E[a][c] = sacc;
						E[c][a] = sacc;
						if (a == c) trg += sacc;
					}
				oct_jacobi_n<6>(E, Ue, lane);
			}
			STAMP(17);
			double tu[6];  // J_u m: the rank-one bounded-inertia form needs it (rows of J_u = U^T J are used on the fly, not kept: 84 registers)
			UNR for (int c = 0; c < 6; c++) {
				double s2 = 0.0, own = 0.0, tacc = 0.0;
				UNR for (int j = 0; j < N; j++) {
					double acc = 0.0;
					UNR for (int i = 0; i < 6; i++) acc = fma(Ue[i][c], Jf[i][j], acc);
					s2 = fma(acc, acc, s2);
					own = (j == rr) ? acc : own;
					tacc = fma(acc, mcol[j], tacc);
				}
				tu[c] = tacc;
				sig[c] = sqrt(s2);  // singular values from the rows themselves: accurate down to ~1e-10 sigma_max (the Gram eigenvalue is not)
				smax = fmax(smax, sig[c]);
				ju[c] = own;
			}
			STAMP(18);
			int ns = 0, cstar = 0;
			double sstar = -1.0;
			bool tiny = false;
			UNR for (int c = 0; c < 6; c++) {
				keep[c] = sig[c] >= mf.s_max * smax;
				ns += keep[c] ? 1 : 0;
				if (!keep[c] && sig[c] > sstar) {
					sstar = sig[c];
					cstar = c;
				}
				tiny = tiny || (!keep[c] && !(sig[c] >= 1e-10 * smax));
			}
			const bool finite_state = (trg - trg == 0.0);  // a NaN or infinite state / goal never passes for "fully singular": refused
			const bool fully = finite_state && !(smax >= mf.s_abs_tol);             // fully singular task: passed through (:149-151, 316-317)
			const bool blend = singular && finite_state && !fully && ns < 6 && !tiny;  // (tiny: the direction of V_s is not resolved: refused)
			const bool clear = singular && finite_state && !fully && ns == 6;        // the eigen-solve overrules a borderline certificate: the ordinary result stands
			if (clear) singular = false;
			if (__any(singular && fully)) {
				// zero task torques, N = N_prec = I: the full joint task behind sees the whole joint space -- U = I, Lambda = M,
				// Lambda_mod = M_BIE / M / I (JointTask.cpp:241-267): tau_r = sum_j M_rj a_j + Mmod_rj f_j
				double acc = 0.0;
				const double af = jg_ddq, ff = fi;  // tau_prec = 0: no disturbance compensation
				UNR for (int j = 0; j < N; j++) {
					const int src = octl_src(lane, j);
					const double aj = __shfl(af, src), fj = __shfl(ff, src);
					const double mrj = (j <= rr) ? smA.M[rr][j] : smA.M[j][rr];
					double mmod = mrj;
					if (jt_dec == DEC_IMPEDANCE) mmod = (j == rr) ? 1.0 : 0.0;
					else if (jt_dec == DEC_BIE) mmod = (j == rr) ? fmax(mrj, jt.bie_threshold) : mrj;
					acc = fma(mrj, aj, fma(mmod, fj, acc));
				}
				if (singular && fully) {
					tvo = acc;
					ovr = true;
					singular = false;
					truncated = true;
				}
			}
			STAMP(19);
			if (__any(blend)) {
				ShState* st = mf.sh + b;
				const int cycle = P.task_cycle[0];
				// ---- handler state of the previous cycle (void unless that cycle was blended too, :238-244), entering posture (:232-235)
				const bool fresh = st->last_cycle != cycle - 1;
				int ntypes = fresh ? 0 : st->ntypes, c1 = fresh ? 0 : st->c1, c2 = fresh ? 0 : st->c2, head = fresh ? 0 : st->head, hsize = fresh ? 0 : st->size;
				double q_prior = st->q_prior[rr];
				if (ntypes == 0 || c2 > c1) q_prior = q_r;
				// ---- classification (:246-294): motion of the control frame along u_c at q +- 5 v_c, the larger of the two
				// (the unperturbed control frame is evaluated again here, the same way, rather than kept alive across the ordinary path)
				double pw0[3], Rc0[9];
				{
					double T0[12], Rb[12];
					oct_fk_frame(md, rr, r, q_r, T0);
					const int srcb = octl_src(lane, mf.body);
					UNR for (int e2 = 0; e2 < 12; e2++) Rb[e2] = __shfl(T0[e2], srcb);
					oct_mat3_vec(Rb, mf.pos, pw0);
					UNR for (int i = 0; i < 3; i++) pw0[i] += Rb[9 + i];
					UNR for (int i = 0; i < 3; i++)
						UNR for (int j = 0; j < 3; j++) Rc0[3 * i + j] = Rb[3 * i] * mf.rot[j] + Rb[3 * i + 1] * mf.rot[3 + j] + Rb[3 * i + 2] * mf.rot[6 + j];
				}
				STAMP(20);
				// one pass per singular direction an instance has (its k-th one, whichever eigen-index that is: usually a single pass), not one
				// per eigen-index that is singular somewhere in the wavefront
				bool any1 = false;
				for (int k = 0; k < 6; k++) {
					if (!__any(blend && k < 6 - ns)) break;  // wave-uniform
					double vnum = 0.0, sk = 1.0, uk[6] = {0, 0, 0, 0, 0, 0};
					int cnt = 0;
					UNR for (int c = 0; c < 6; c++) {
						const bool sel = !keep[c] && cnt == k;
						vnum = sel ? ju[c] : vnum;
						sk = sel ? sig[c] : sk;
						UNR for (int i = 0; i < 6; i++) uk[i] = sel ? Ue[i][c] : uk[i];
						cnt += keep[c] ? 0 : 1;
					}
					const double vc = vnum * oct_rcp(sk);
					double motion = 0.0;
					for (int sg = 0; sg < 2; sg++) {
						double T1w[12], Rb[12], x1[3], R1[9];
						oct_fk_frame(md, rr, r, q_r + (sg ? -5.0 : 5.0) * vc, T1w);
						const int srcb = octl_src(lane, mf.body);
						UNR for (int e2 = 0; e2 < 12; e2++) Rb[e2] = __shfl(T1w[e2], srcb);
						oct_mat3_vec(Rb, mf.pos, x1);
						UNR for (int i = 0; i < 3; i++) x1[i] += Rb[9 + i];
						UNR for (int i = 0; i < 3; i++)
							UNR for (int j = 0; j < 3; j++) R1[3 * i + j] = Rb[3 * i] * mf.rot[j] + Rb[3 * i + 1] * mf.rot[3 + j] + Rb[3 * i + 2] * mf.rot[6 + j];
						double w6[6] = {x1[0] - pw0[0], x1[1] - pw0[1], x1[2] - pw0[2], 0.0, 0.0, 0.0};
						UNR for (int col = 0; col < 3; col++) {  // orientationError(R1, Rc)
							const double a0 = Rc0[col], a1 = Rc0[3 + col], a2 = Rc0[6 + col], b0 = R1[col], b1 = R1[3 + col], b2 = R1[6 + col];
							w6[3] -= 0.5 * (a1 * b2 - a2 * b1);
							w6[4] -= 0.5 * (a2 * b0 - a0 * b2);
							w6[5] -= 0.5 * (a0 * b1 - a1 * b0);
						}
						double d = 0.0;
						UNR for (int i = 0; i < 6; i++) d = fma(w6[i], uk[i], d);
						motion = fmax(motion, fabs(d));
					}
					any1 = any1 || (k < 6 - ns && motion > 0.5);  // _type_1_tol
				}
				STAMP(21);
				if (hsize == SH_BUFFER) {  // :285-292
					if (st->hist[head] == 1) c1--;
					else c2--;
					head = (head + 1) % SH_BUFFER;
					hsize--;
				}
				const int slot = (head + hsize) % SH_BUFFER;
				hsize++;
				if (any1) c1++;
				else c2++;
				const bool type1 = (c1 > c2) || mf.sh_force_type1;  // :329
				// type-2 direction memory (:339-345): joints that take part in the first singular direction flip near their limits
				unsigned dir_neg = st->dir_neg;
				{
					double jstar = 0.0;
					UNR for (int c = 0; c < 6; c++) jstar = (c == cstar) ? ju[c] : jstar;
					const bool part = act && jstar != 0.0;
					const bool setb = part && fabs(q_r - md.q_upper[rr]) < 5.0 * M_PI / 180.0;
					const bool clrb = part && !setb && fabs(q_r - md.q_lower[rr]) < 5.0 * M_PI / 180.0;
					const unsigned sets = octl_votes(__ballot(setb), lane) & 0x7fu, clrs = octl_votes(__ballot(clrb), lane) & 0x7fu;
					if (!type1) dir_neg = (dir_neg | sets) & ~clrs;
				}
				if (blend && live) {
					if (act) st->q_prior[r] = q_prior;
					if (r == 0) {
						st->hist[slot] = any1 ? 1 : 2;
						st->ntypes = 6 - ns;
						st->c1 = c1;
						st->c2 = c2;
						st->head = head;
						st->size = hsize;
						st->dir_neg = dir_neg;
						st->last_cycle = cycle;
					}
				}
				STAMP(22);
				// ---- A_u = U^T A U and its bounded-inertia version (A from LDS: the ordinary path left it there)
				double Au[6][6], Amu[6][6];
				{
					double Af[6][6], W[6][6];
					UNR for (int i = 0; i < 6; i++)
						UNR for (int j = 0; j <= i; j++) {
							Af[i][j] = smA.A[i][j];
							Af[j][i] = Af[i][j];
						}
					UNR for (int i = 0; i < 6; i++)
						UNR for (int c = 0; c < 6; c++) {
							double acc = 0.0;
							UNR for (int k = 0; k < 6; k++) acc = fma(Af[i][k], Ue[k][c], acc);
							W[i][c] = acc;
						}
					UNR for (int c = 0; c < 6; c++)
						UNR for (int d = 0; d <= c; d++) {
							double acc = 0.0;
							UNR for (int i = 0; i < 6; i++) acc = fma(Ue[i][c], W[i][d], acc);
							Au[c][d] = acc;
							Au[d][c] = acc;
						}
					if (mf_dec == DEC_BIE && general_bie) {
						UNR for (int i = 0; i < 6; i++)
							UNR for (int j = 0; j <= i; j++) {
								Af[i][j] = smA.Am[i][j];
								Af[j][i] = Af[i][j];
							}
						UNR for (int i = 0; i < 6; i++)
							UNR for (int c = 0; c < 6; c++) {
								double acc = 0.0;
								UNR for (int k = 0; k < 6; k++) acc = fma(Af[i][k], Ue[k][c], acc);
								W[i][c] = acc;
							}
						UNR for (int c = 0; c < 6; c++)
							UNR for (int d = 0; d <= c; d++) {
								double acc = 0.0;
								UNR for (int i = 0; i < 6; i++) acc = fma(Ue[i][c], W[i][d], acc);
								Amu[c][d] = acc;
								Amu[d][c] = acc;
							}
					} else {
						const double bb = (mf_dec == DEC_BIE) ? beta : 0.0;  // rank-one form: A_mod = A - beta t t^T (beta = 0: nothing clamped)
						UNR for (int c = 0; c < 6; c++)
							UNR for (int d = 0; d < 6; d++) Amu[c][d] = fma(-bb * tu[c], tu[d], Au[c][d]);
					}
				}
				STAMP(23);
				// ---- block-diagonal factors: Lambda_ns / Lambda_s and their modified versions in one Cholesky each
				double LB[6][6], dB[6], LM[6][6], dM[6];
				UNR for (int c = 0; c < 6; c++)
					UNR for (int d = 0; d <= c; d++) {
						const bool same = keep[c] == keep[d];
						LB[c][d] = same ? Au[c][d] : 0.0;
						LM[c][d] = same ? Amu[c][d] : 0.0;
					}
				oct_cholesky<6>(LB, dB);
				oct_cholesky<6>(LM, dM);
				STAMP(24);
				// ---- task forces in the eigenbasis, tau_ns and the clamped tau_s (:321-322, 358-366)
				double fu[6], ffu[6], gu[6], fn2 = 0.0;
				UNR for (int c = 0; c < 6; c++) {
					double a = 0.0, bq = 0.0;
					UNR for (int i = 0; i < 6; i++) {
						a = fma(Ue[i][c], Fum6[i], a);
						bq = fma(Ue[i][c], Ff6[i], bq);
					}
					fu[c] = a;
					ffu[c] = bq;
				}
				UNR for (int i = 0; i < 6; i++) fn2 = fma(Fum6[i] + Ff6[i], Fum6[i] + Ff6[i], fn2);
				if (mf_dec == DEC_IMPEDANCE) {
					UNR for (int c = 0; c < 6; c++) gu[c] = fu[c] + ffu[c];
				} else {
					oct_solve<6>(LM, dM, fu, gu);
					UNR for (int c = 0; c < 6; c++) gu[c] += ffu[c];
				}
				double tns = 0.0, ts = 0.0;
				UNR for (int c = 0; c < 6; c++) {
					tns = fma(keep[c] ? ju[c] : 0.0, gu[c], tns);
					ts = fma(keep[c] ? 0.0 : ju[c], gu[c], ts);
				}
				{
					const double lim = md.effort[rr];
					ts = (ts != ts) ? 0.0 : fmin(fmax(ts, -lim), lim);
				}
				double sstar_v = 0.0, fstar = 0.0;
				UNR for (int c = 0; c < 6; c++) {
					sstar_v = (c == cstar) ? sig[c] : sstar_v;
					fstar = (c == cstar) ? fu[c] + ffu[c] : fstar;
				}
				const double alpha = fmin(fmax((sstar_v * oct_rcp(smax) - mf.s_min) * oct_rcp(mf.s_max - mf.s_min), 0.0), 1.0);  // :114-115
				const double fTd = fstar * oct_rsqrt(fn2);  // :346-347
				STAMP(25);
				// ---- joint strategy: posture Jacobian rows (own column), its op-space inertia from the Schur complement of A_u
				double X[6][6], jpp[6], Pj[6][6], Pm[6][6];
				UNR for (int c = 0; c < 6; c++) {
					double rhs[6];
					UNR for (int d = 0; d < 6; d++) rhs[d] = (keep[d] && !keep[c]) ? Au[d][c] : 0.0;
					oct_solve<6>(LB, dB, rhs, X[c]);  // x_c = Lambda_ns A_u[K, c], supported on K (zero for c in K)
				}
				UNR for (int c = 0; c < 6; c++) {
					const double isg = keep[c] ? 0.0 : oct_rcp(sig[c]);
					double acc = ju[c];
					UNR for (int d = 0; d < 6; d++) acc = fma(-X[c][d], ju[d], acc);
					jpp[c] = acc * isg;
				}
				UNR for (int c = 0; c < 6; c++) {
					double yc[6];  // y_c = A_mod,u x_c
					UNR for (int d2 = 0; d2 < 6; d2++) {
						double acc = 0.0;
						UNR for (int d = 0; d < 6; d++) acc = fma(Amu[d2][d], X[c][d], acc);
						yc[d2] = acc;
					}
					UNR for (int e2 = 0; e2 <= c; e2++) {
						const bool both = !keep[c] && !keep[e2];
						const double isg = both ? oct_rcp(sig[c]) * oct_rcp(sig[e2]) : 0.0;
						double sch = Au[c][e2], schm = Amu[c][e2] - yc[e2], xy = 0.0, ye = 0.0;
						UNR for (int d = 0; d < 6; d++) {
							sch = fma(-X[c][d], Au[d][e2], sch);
							xy = fma(X[e2][d], yc[d], xy);
							ye = fma(Amu[c][d], X[e2][d], ye);  // (y_e)_c
						}
						schm = schm - ye + xy;
						Pj[c][e2] = both ? sch * isg : ((c == e2) ? 1.0 : 0.0);
						Pm[c][e2] = both ? schm * isg : ((c == e2) ? 1.0 : 0.0);
					}
				}
				if (mf_dec == DEC_FULL) {
					UNR for (int c = 0; c < 6; c++)
						UNR for (int e2 = 0; e2 <= c; e2++) Pm[c][e2] = Pj[c][e2];
				}
				STAMP(26);
				double dPm[6];
				oct_cholesky<6>(Pm, dPm);
				// unit torques (:333-336 type 1: hold the entering posture; :346-354 type 2) and V_s^T of them (sums over the lanes)
				const double u1 = type1 ? (-mf.sh_kp1 * (q_r - q_prior) - mf.sh_kv1 * dq_r)
										: ((((dir_neg >> rr) & 1u) ? -1.0 : 1.0) * (fabs(fTd) * 1e-2 * md.effort[rr]));
				const double u2 = -mf.sh_kv2 * dq_r;
				double w1[6], w2[6], lw[6];
				UNR for (int c = 0; c < 6; c++) {
					const double vc = (keep[c] || !act) ? 0.0 : ju[c] * oct_rcp(sig[c]);
					w1[c] = octl_sum(vc * u1);
					w2[c] = type1 ? 0.0 : octl_sum(vc * u2);
				}
				double tjs = 0.0, wsel[6];
				UNR for (int c = 0; c < 6; c++) wsel[c] = type1 ? w1[c] : w2[c];  // (element-wise: never a run-time choice between two private arrays)
				if (mf_dec == DEC_IMPEDANCE) {
					UNR for (int c = 0; c < 6; c++) lw[c] = wsel[c];
				} else {
					oct_solve<6>(Pm, dPm, wsel, lw);
				}
				UNR for (int c = 0; c < 6; c++) tjs = fma(jpp[c], type1 ? lw[c] : (w1[c] + lw[c]), tjs);
				const double tmf = (mf_dec == DEC_IMPEDANCE) ? tns : (tns + alpha * ts + (1.0 - alpha) * tjs);  // :311-313, :367
				STAMP(27);
				// ---- the joint task behind: N = N_js N_ns projects onto null(J) along M^-1 range(J^T), so N = z z^T M / (z^T M z) again;
				// z from the Euclidean projector I - sum_c v_c v_c^T (pivot row: the lane with the largest diagonal entry)
				double zn = act ? 1.0 : -1.0;
				double vown[6];
				UNR for (int c = 0; c < 6; c++) {
					vown[c] = ju[c] * oct_rcp(sig[c]);
					zn = fma(-vown[c], vown[c], zn);
				}
				double bestw = zn;
				int jbw = r;
	#define OCT_ARGMAX_STEP(CTRL)                                                                                   \
		{                                                                                                           \
		const double ob = oct_dpp<CTRL>(bestw);                                                                 \
		const int oj = __builtin_amdgcn_update_dpp(0, jbw, CTRL, 0xF, 0xF, true);                               \
		const bool take = ob > bestw || (ob == bestw && oj < jbw);                                              \
		bestw = take ? ob : bestw;                                                                              \
		jbw = take ? oj : jbw;                                                                                  \
		}
				OCT_ARGMAX_STEP(OCT_BFLY0)
				OCT_ARGMAX_STEP(OCT_BFLY1)
				OCT_ARGMAX_STEP(OCT_BFLY2)
	#undef OCT_ARGMAX_STEP
				const int srcp = octl_src(lane, jbw);
				double zb = (r == jbw) ? 1.0 : 0.0;
				UNR for (int c = 0; c < 6; c++) zb = fma(-vown[c], __shfl(vown[c], srcp), zb);
				zb = act ? zb : 0.0;
				double mz = 0.0, wq = 0.0;
				UNR for (int j = 0; j < N; j++) {
					const int srcj = octl_src(lane, j);
					const double zj = __shfl(zb, srcj), tj = __shfl(tmf, srcj);
					const double mrj = (j <= rr) ? smA.M[rr][j] : smA.M[j][rr];
					mz = fma(mrj, zj, mz);   // (M z)_rr
					wq = fma(mi[j], tj, wq);  // (M^-1 tau_prec)_rr
				}
				const double aib = jg_ddq - wq;
				const double zzb = octl_sum(zb * zb), sb = octl_sum(act ? zb * mz : 0.0);
				const double zab = octl_sum(zb * aib), zfb = octl_sum(zb * fi);
				double rhob = 1.0;  // rho |z|^2 in the notation above: tau = (M z)_r / |z|^2 (z.a + rho' z.f), rho' = Lambda_mod / Lambda
				if (jt_dec == DEC_IMPEDANCE) {
					rhob = zzb * oct_rcp(sb);
				} else if (jt_dec == DEC_BIE) {
					if (general_bie) {
						double mbv = 0.0;
						UNR for (int j = 0; j < N; j++) mbv = fma(mb[j], __shfl(mz, octl_src(lane, j)), mbv);
						const double qb = octl_sum(act ? mz * mbv : 0.0);
						rhob = sb * oct_rcp(qb);
					} else {
						const double ze = octl_sum((act && r == eclamp) ? zb : 0.0);
						rhob = sb * oct_rcp(fma(-beta * ze, ze, sb));
					}
				}
				STAMP(28);
				const double tvb = fma(mz * oct_rcp(zzb), fma(rhob, zfb, zab), tmf);
				if (blend) {
					tvo = tvb;
					ovr = true;
					singular = false;
					blended_i = true;
				}
			}
		}
		if (DUO) {
			// second workgroup barrier: B hands over the nullspace direction (z', v = M z up to scale, |z'|^2, z'.v) and exits
			if (ROLE == 2) {
				sm.vec[7][r] = zr;
				sm.vec[8][r] = v_own;
				sm.vec[1][r] = tvo;
				sm.vec[2][r] = (ovr ? 1.0 : 0.0) + (singular ? 2.0 : 0.0) + (truncated ? 4.0 : 0.0) + (blended_i ? 8.0 : 0.0) + (lost ? 16.0 : 0.0);
				if (r == 0) {
					sm.vec[3][0] = zz;
					sm.vec[3][1] = zv;
					UNR for (int j = 0; j < N; j++) sm.vec[4][j] = v[j];
				}
			}
			STAMP(31);
			__syncthreads();
			if (ROLE == 2) return false;
			zr = smB.vec[7][r];
			v_own = smB.vec[8][r];
			zz = smB.vec[3][0];
			zv = smB.vec[3][1];
			{
				tvo = smB.vec[1][r];
				const int fl = (int)smB.vec[2][r];
				ovr = (fl & 1) != 0;
				singular = (fl & 2) != 0;
				truncated = (fl & 4) != 0;
				blended_i = (fl & 8) != 0;
				lost = lost || (fl & 16) != 0;
			}
		}
		if (ROLE == 1) {
		const double ai = jg_ddq - wr;
		const double za = octl_sum(zr * ai);
		const double zf = octl_sum(zr * fi);
		double rho = 1.0;
		if (jt_dec == DEC_IMPEDANCE) {
			rho = zz * oct_rcp(zv);
		} else if (jt_dec == DEC_BIE) {
			if (general_bie) {
				if (DUO) {
					UNR for (int j = 0; j < N; j++) v[j] = smB.vec[4][j];
				}
				double mbv = 0.0;
				UNR for (int j = 0; j < N; j++) mbv = fma(mb[j], v[j], mbv);  // (M_BIE^-1 v)_rr
				const double qb = octl_sum(act ? v_own * mbv : 0.0);
				rho = zv * oct_rcp(qb);
			} else {
				const double ze = octl_sum((act && r == eclamp) ? zr : 0.0);
				rho = zv * oct_rcp(fma(-beta * ze, ze, zv));
			}
		}
		tv = fma(v_own * oct_rcp(zz), fma(rho, zf, za), tau_r);  // tau += Jh^T g
		}
		if (ovr) tv = tvo;  // the blended strategies / the pass-through of a fully singular task decided this instance's torque
		if (BROWS && lost) singular = true;

	} else {
	{
		double g[6], t3[6], jc[6];
		UNR for (int a = 0; a < 6; a++) {
			g[a] = sm.vec[2][a];
			jc[a] = (ROLE == 1 ? smB : sm).J[a][rr];
		}
		double s = 0.0;
		UNR for (int a = 0; a < 6; a++) s = fma(jc[a], g[a], s);
		tau_r = s;
		UNR for (int a = 0; a < 6; a++) {
			double acc = 0.0;
			UNR for (int c = 0; c < 6; c++) acc = fma(sm.Lam[a][c], jc[c], acc);
			t3[a] = acc;
		}
		UNR for (int i = 0; i < N; i++) {
			double acc = (i == rr) ? 1.0 : 0.0;
			UNR for (int a = 0; a < 6; a++) acc = fma(-sm.T1[a][i], t3[a], acc);
			cj[i] = acc;
			cn = fma(acc, acc, cn);
		}
		UNR for (int i = 0; i < N; i++) sm.N1[r][i] = cj[i];  // transposed storage: row r = column r of N_1
		sm.vec[1][r] = tau_r;
	}
	STAMP(11);
	// ---------------------------------------------------------------- JointTask, general path (GJ): any selection matrix with m <= 4 rows behind any
	// motion-force task.  Lane j holds column j of Jp = S N_1 (m x 7).  Left singular vectors / squared singular values of Jp = eigen-pairs
	// of its 4 x 4 Gram matrix (sums over the lanes by DPP), by an unrolled Jacobi solve in registers; matrixRangeBasis keeps the directions
	// with sigma_i / sigma_0 >= 1e-3 (JointTask.cpp:233); the rest of the task algebra is r x r with dropped directions padded.
	if (GJ == 1) {
		double cp[4];
		UNR for (int i = 0; i < 4; i++) {
			double sc = 0.0;
			UNR for (int l = 0; l < N; l++) sc = fma(jt.S[i * N + l], cj[l], sc);  // rows >= m of the stored S are zero
			cp[i] = act ? sc : 0.0;
		}
		double Gp[4][4], V4[4][4], trp = 0.0;
		UNR for (int i = 0; i < 4; i++)
			UNR for (int k = 0; k <= i; k++) {
				const double gs = octl_sum(cp[i] * cp[k]);
				Gp[i][k] = gs;
				Gp[k][i] = gs;
				if (i == k) trp += gs;
			}
		// Usual case: Jp has full row rank, matrixRangeBasis returns the identity (sai-model: task_dof == rows) and no eigen-solve is needed.
		// Certificate (the one of the motion-force branch predicate): lambda_max <= trace, so a positive definite  Gp - 1e-6 trace I  on
		// the m task rows proves sigma_min / sigma_max > 1e-3; trace / m >= 1e-6 proves sigma_max >= 1e-3.  Only wavefronts that hold an
		// instance it cannot decide run the register Jacobi solve (measured, config 3: 19 k of 40 k clocks per wavefront were this solve).
		bool full_rank = trp * 0.25 >= 1e-6;
		{
			double Gs[4][4];
			const double sh = 1e-6 * trp;
			UNR for (int i = 0; i < 4; i++)
				UNR for (int k = 0; k <= i; k++) Gs[i][k] = Gp[i][k];
			UNR for (int i = 0; i < 4; i++) Gs[i][i] = (i < mj) ? Gs[i][i] - sh : 1.0;
			UNR for (int kk = 0; kk < 4; kk++) {
				const double d = Gs[kk][kk];
				full_rank = full_rank && (d > 1e-13 * trp);
				const double id = oct_rcp(d);
				UNR for (int i = kk + 1; i < 4; i++) {
					const double lik = Gs[i][kk] * id;
					UNR for (int k = kk + 1; k <= i; k++) Gs[i][k] = fma(-lik, Gs[k][kk], Gs[i][k]);
				}
			}
		}
#if defined(SAIP_OCT_FORCE_EXACT)
		full_rank = false;
#endif
		bool keep[4];
		double jh[4];
		UNR for (int i = 0; i < 4; i++)
			UNR for (int k = 0; k < 4; k++) V4[i][k] = (i == k) ? 1.0 : 0.0;
		UNR for (int c = 0; c < 4; c++) {
			keep[c] = c < mj;
			jh[c] = keep[c] ? cp[c] : 0.0;
		}
		if (__any(!full_rank)) {
			double Ve[4][4];
			oct_jacobi4(Gp, Ve, lane);
			double lmaxp = 0.0;
			UNR for (int c = 0; c < 4; c++) lmaxp = fmax(lmaxp, Gp[c][c]);
			const bool any_range = (sqrt(fmax(trp, 0.0)) >= 1e-3) && (sqrt(lmaxp) >= 1e-3);
			UNR for (int c = 0; c < 4; c++) {
				const bool kc = any_range && (sqrt(fmax(Gp[c][c], 0.0) / lmaxp) >= 1e-3);
				double sj = 0.0;
				UNR for (int i = 0; i < 4; i++) sj = fma(Ve[i][c], cp[i], sj);
				if (!full_rank) {
					keep[c] = kc;
					jh[c] = kc ? sj : 0.0;  // own column of Jh = U^T Jp
					UNR for (int i = 0; i < 4; i++) V4[i][c] = Ve[i][c];
				}
			}
		}
		// gather Jh, q, the joint-task goal rows and w = M^-1 tau_prec for the whole instance
		double tauv[N];
		UNR for (int j = 0; j < N; j++) tauv[j] = sm.vec[1][j];
		double wr = 0.0;
		UNR for (int l = 0; l < N; l++) wr = fma(mi[l], tauv[l], wr);
		UNR for (int c = 0; c < 4; c++) sm.T1[c][r] = jh[c];
		sm.vec[4][r] = wr;
		sm.vec[6][r] = q_r;
		sm.vec[7][r] = jg_q;
		sm.vec[8][r] = jg_dq;
		sm.vec[3][r] = jg_ddq;
		oct_sync<DUO>();
		double Jh[4][N];
		UNR for (int c = 0; c < 4; c++)
			UNR for (int j = 0; j < N; j++) Jh[c][j] = sm.T1[c][j];
		double t1v[4], A4[4][4], Lam4[4][4], Lmod4[4][4];
		UNR for (int c = 0; c < 4; c++) {
			double st = 0.0;
			UNR for (int l = 0; l < N; l++) st = fma(Jh[c][l], mi[l], st);  // (Jh M^-1)[c][rr]: M^-1 symmetric
			t1v[c] = act ? st : 0.0;
		}
		UNR for (int c = 0; c < 4; c++)
			UNR for (int k = 0; k <= c; k++) {
				const double gs = octl_sum(t1v[c] * jh[k]);
				A4[c][k] = (c == k && !keep[c]) ? 1.0 : gs;
			}
		oct_spd_inverse4(A4, Lam4);
		if (jt_dec == DEC_FULL) {
			UNR for (int c = 0; c < 4; c++)
				UNR for (int k = 0; k < 4; k++) Lmod4[c][k] = Lam4[c][k];
		} else if (jt_dec == DEC_IMPEDANCE) {
			UNR for (int c = 0; c < 4; c++)
				UNR for (int k = 0; k < 4; k++) Lmod4[c][k] = (c == k) ? 1.0 : 0.0;
		} else if (general_bie) {
			double tbv[4];
			UNR for (int c = 0; c < 4; c++) {
				double st = 0.0;
				UNR for (int l = 0; l < N; l++) st = fma(Jh[c][l], mb[l], st);
				tbv[c] = act ? st : 0.0;
			}
			UNR for (int c = 0; c < 4; c++)
				UNR for (int k = 0; k <= c; k++) {
					const double gs = octl_sum(tbv[c] * jh[k]);
					A4[c][k] = (c == k && !keep[c]) ? 1.0 : gs;
				}
			oct_spd_inverse4(A4, Lmod4);
		} else {  // rank-one bounded-inertia form: Lambda_mod = Lambda + gamma (Lambda t)(Lambda t)^T, t = Jh m
			double t[4], lt[4], qq = 0.0;
			UNR for (int c = 0; c < 4; c++) {
				double st = 0.0;
				UNR for (int l = 0; l < N; l++) st = fma(Jh[c][l], mcol[l], st);
				t[c] = st;
			}
			UNR for (int c = 0; c < 4; c++) {
				double sl = 0.0;
				UNR for (int k = 0; k < 4; k++) sl = fma(Lam4[c][k], t[k], sl);
				lt[c] = sl;
				qq = fma(t[c], sl, qq);
			}
			const double gamma = beta * oct_rcp(fma(-beta, qq, 1.0));
			UNR for (int c = 0; c < 4; c++)
				UNR for (int k = 0; k < 4; k++) Lmod4[c][k] = fma(gamma * lt[c], lt[k], Lam4[c][k]);
		}
		// control law of the (at most four) task rows, every lane alike (JointTask.cpp:285-356)
		double qa[N], dqa[N], wa[N];
		UNR for (int j = 0; j < N; j++) {
			qa[j] = sm.vec[6][j];
			dqa[j] = sm.vec[0][j];
			wa[j] = sm.vec[4][j];
		}
		const bool track = (jt.has_ki || P.integ_always);
		double a1[4] = {0, 0, 0, 0}, b1[4] = {0, 0, 0, 0};
		UNR for (int i = 0; i < 4; i++) {
			double cur = 0.0, vel = 0.0, sw = 0.0;
			UNR for (int l = 0; l < N; l++) {
				const double sil = jt.S[i * N + l];
				cur = fma(sil, qa[l], cur);
				vel = fma(sil, dqa[l], vel);
				sw = fma(sil, wa[l], sw);
			}
			const bool row = i < mj;
			const double gq = row ? sm.vec[7][i] : 0.0, gdq = row ? sm.vec[8][i] : 0.0, gddq = row ? sm.vec[3][i] : 0.0;
			const double e = cur - gq;
			double ie = (track && row) ? sm.ist[6 + i] : 0.0;
			ie += e * jt.dt;  // :323-324
			jt_ie_new = (r == i) ? ie : jt_ie_new;  // lane i keeps row i for the epilogue
			double fi;
			if (jt.vel_sat) {  // :327-341
				double vdes = -jt.kp[i] * jt.kvinv[i] * e - jt.ki[i] * jt.kvinv[i] * ie;
				vdes = fmin(fmax(vdes, -jt.sat[i]), jt.sat[i]);
				fi = -jt.kv[i] * (vel - vdes);
			} else {
				fi = -jt.kp[i] * e - jt.kv[i] * (vel - gdq) - jt.ki[i] * ie;  // :342-345
			}
			const double ai = gddq - sw;
			UNR for (int c = 0; c < 4; c++) {
				a1[c] = fma(V4[i][c], row ? ai : 0.0, a1[c]);
				b1[c] = fma(V4[i][c], row ? fi : 0.0, b1[c]);
			}
		}
		double tj = tau_r;
		UNR for (int c = 0; c < 4; c++) {
			double gc = 0.0;
			UNR for (int k = 0; k < 4; k++) gc = fma(Lam4[c][k], a1[k], fma(Lmod4[c][k], b1[k], gc));  // :348-351
			tj = fma(jh[c], keep[c] ? gc : 0.0, tj);  // tau += Jh^T g
		}
		tv = tj;
	} else if (GJ == 2) {
		// ------------------------------------------------------------ full joint task (S = I) behind a motion-force task of rank k < 6: Jp = N_1 has
		// rank 7 - k (2..5).  Column-pivoted Gram-Schmidt over the instance's lanes (lane j owns column j): arg-max of the column norms by
		// a DPP butterfly carrying the index, the pivot column travels by ds_bpermute, the deflation coefficients are the rows of
		// Jh = U^T Jp (the same clean-gap rule as the lane kernel; an ambiguous gap is reported as status 1, never silently truncated).
		const int rmax = trunc_mode ? 5 : N - kmf;  // wave-uniform bound; a reduced task leaves rank 7 - ns <= 5, found by the clean-gap rule
		double Wc[N], jh[5], uown[5];
		UNR for (int i = 0; i < N; i++) Wc[i] = act ? cj[i] : 0.0;
		UNR for (int s5 = 0; s5 < 5; s5++) jh[s5] = uown[s5] = 0.0;
		double c0 = 1.0;
		bool going = true, bad = false;
		int rank = 0;
		UNR for (int s5 = 0; s5 <= 5; s5++) {
			if (s5 <= rmax) {
				double cnw = 0.0;
				UNR for (int i = 0; i < N; i++) cnw = fma(Wc[i], Wc[i], cnw);
				double bestw = act ? cnw : -1.0;
				int jbw = r;
#define OCT_ARGMAX_STEP(CTRL)                                                                                   \
	{                                                                                                           \
		const double ob = oct_dpp<CTRL>(bestw);                                                                 \
		const int oj = __builtin_amdgcn_update_dpp(0, jbw, CTRL, 0xF, 0xF, true);                               \
		const bool take = ob > bestw || (ob == bestw && oj < jbw);                                              \
		bestw = take ? ob : bestw;                                                                              \
		jbw = take ? oj : jbw;                                                                                  \
	}
				OCT_ARGMAX_STEP(OCT_BFLY0)
				OCT_ARGMAX_STEP(OCT_BFLY1)
				OCT_ARGMAX_STEP(OCT_BFLY2)
#undef OCT_ARGMAX_STEP
				if (s5 == 0) {
					const double frob2 = octl_sum(act ? cnw : 0.0);
					c0 = bestw;
					if (frob2 < 1e-6) going = false;  // ||Jp||_F < 1e-3: empty range
					else if (bestw < 1e-5) { going = false; bad = true; }
				} else if (going) {
					const double ratio = bestw / c0;
					if (ratio < 1e-20) going = false;                         // numerically exact rank below the bound
					else if (ratio < 1e-4 || s5 == rmax) { going = false; bad = true; }  // ambiguous gap, or more directions than 7 - k
				}
				if (s5 < rmax && s5 < 5) {
					const int src = octl_src(lane, jbw);
					const double inv = going ? oct_rsqrt(bestw) : 0.0;
					double u[N], dd = 0.0;
					UNR for (int i = 0; i < N; i++) {
						u[i] = __shfl(Wc[i], src) * inv;
						dd = fma(u[i], Wc[i], dd);
					}
					UNR for (int i = 0; i < N; i++) Wc[i] = fma(-u[i], dd, Wc[i]);
					double uo = 0.0;
					UNR for (int i = 0; i < N; i++) uo = (i == rr) ? u[i] : uo;
					jh[s5 < 5 ? s5 : 0] = act ? dd : 0.0;
					uown[s5 < 5 ? s5 : 0] = act ? uo : 0.0;
					if (going) rank = s5 + 1;
				}
			}
		}
		if (bad) singular = true;
		// Lambda = (Jh M^-1 Jh^T)^-1 (rank x rank, padded to 5): gather Jh, own column of T1, sums over the lanes
		double tauv[N];
		UNR for (int j = 0; j < N; j++) tauv[j] = sm.vec[1][j];
		UNR for (int c = 0; c < 5; c++) sm.T1[c][r] = jh[c];
		oct_sync<DUO>();
		double Jh[5][N];
		UNR for (int c = 0; c < 5; c++)
			UNR for (int j = 0; j < N; j++) Jh[c][j] = sm.T1[c][j];
		double t1v[5], wr = 0.0, A5[5][5], Lam5[5][5], Lmod5[5][5];
		UNR for (int l = 0; l < N; l++) wr = fma(mi[l], tauv[l], wr);  // (M^-1 tau_prec)_rr
		UNR for (int c = 0; c < 5; c++) {
			double st = 0.0;
			UNR for (int l = 0; l < N; l++) st = fma(Jh[c][l], mi[l], st);
			t1v[c] = act ? st : 0.0;
		}
		UNR for (int c = 0; c < 5; c++)
			UNR for (int k5 = 0; k5 <= c; k5++) {
				const double gs = octl_sum(t1v[c] * jh[k5]);
				A5[c][k5] = (c == k5 && c >= rank) ? 1.0 : gs;
			}
		oct_spd_inverse_n<5>(A5, Lam5);
		if (jt_dec == DEC_FULL) {
			UNR for (int c = 0; c < 5; c++)
				UNR for (int k5 = 0; k5 < 5; k5++) Lmod5[c][k5] = Lam5[c][k5];
		} else if (jt_dec == DEC_IMPEDANCE) {
			UNR for (int c = 0; c < 5; c++)
				UNR for (int k5 = 0; k5 < 5; k5++) Lmod5[c][k5] = (c == k5) ? 1.0 : 0.0;
		} else if (general_bie) {
			double tbv[5];
			UNR for (int c = 0; c < 5; c++) {
				double st = 0.0;
				UNR for (int l = 0; l < N; l++) st = fma(Jh[c][l], mb[l], st);
				tbv[c] = act ? st : 0.0;
			}
			UNR for (int c = 0; c < 5; c++)
				UNR for (int k5 = 0; k5 <= c; k5++) {
					const double gs = octl_sum(tbv[c] * jh[k5]);
					A5[c][k5] = (c == k5 && c >= rank) ? 1.0 : gs;
				}
			oct_spd_inverse_n<5>(A5, Lmod5);
		} else {
			double t[5], lt[5], qq = 0.0;
			UNR for (int c = 0; c < 5; c++) {
				double st = 0.0;
				UNR for (int l = 0; l < N; l++) st = fma(Jh[c][l], mcol[l], st);
				t[c] = st;
			}
			UNR for (int c = 0; c < 5; c++) {
				double sl = 0.0;
				UNR for (int k5 = 0; k5 < 5; k5++) sl = fma(Lam5[c][k5], t[k5], sl);
				lt[c] = sl;
				qq = fma(t[c], sl, qq);
			}
			const double gamma = beta * oct_rcp(fma(-beta, qq, 1.0));
			UNR for (int c = 0; c < 5; c++)
				UNR for (int k5 = 0; k5 < 5; k5++) Lmod5[c][k5] = fma(gamma * lt[c], lt[k5], Lam5[c][k5]);
		}
		// (the control law of this lane's joint ran behind M(q): fi_pre); its range coordinates by sums over the lanes
		const double fi = fi_pre;
		const double ai = jg_ddq - wr;
		double a1[5], b1[5];
		UNR for (int c = 0; c < 5; c++) {
			a1[c] = octl_sum(uown[c] * ai);
			b1[c] = octl_sum(uown[c] * fi);
		}
		double tj = tau_r;
		UNR for (int c = 0; c < 5; c++) {
			double gc = 0.0;
			UNR for (int k5 = 0; k5 < 5; k5++) gc = fma(Lam5[c][k5], a1[k5], fma(Lmod5[c][k5], b1[k5], gc));
			tj = fma(jh[c], (c < rank) ? gc : 0.0, tj);
		}
		tv = tj;
	}
	}  // GJ != 0
	STAMP(14);
#if defined(SAIP_OCT_EXP_FORCE_SCRATCH)  // diagnostic build only: does the mere use of scratch memory by a kernel cost launch time?
	if (P.B < 0) {
		volatile double spill[16];
		spill[lane & 15] = tv;
		tv += spill[(lane + 3) & 15];
	}
#endif
	// ---------------------------------------------------------------- post-processing, RobotController.cpp:86-116
	if (P.torque_sat) {
		const double lim = md.effort[rr];
		tv = tv > lim ? lim : (tv < -lim ? -lim : tv);
	}
	if (FULL && P.jla) {
		// joint limit avoidance wrap, RobotController.cpp:96-112: tau = JLA.computeTorques(tau) + N_c^T tau with
		// N_c^T = I - S^T (S M^-1 S^T)^-1 S M^-1 over the joints inside a limit zone (padded to 7 x 7 with identity rows).  The zone logic is
		// per joint = per lane; the rest only runs when some instance of the wavefront touches a zone.
		bool zone;
		const double tj = jla_joint(q_r, dq_r, md.q_lower[rr], md.q_upper[rr], md.vel_limit[rr], md.effort[rr], tv, &zone);
		const bool in_zone = zone && act && !singular;
		if (__any(in_zone)) {
			sm.vec[1][r] = tv;
			sm.vec[3][r] = in_zone ? 1.0 : 0.0;
			oct_sync<DUO>();
			double tall[N], zall[N];
			UNR for (int j = 0; j < N; j++) {
				tall[j] = sm.vec[1][j];
				zall[j] = sm.vec[3][j];
			}
			double sacc = 0.0;
			UNR for (int j = 0; j < N; j++) sacc = fma(mi[j], tall[j], sacc);  // (M^-1 tau)_rr
			sm.vec[4][r] = in_zone ? sacc : 0.0;
			UNR for (int j = 0; j < N; j++) sm.N1[r][j] = (in_zone && zall[j] != 0.0) ? mi[j] : ((j == rr) ? 1.0 : 0.0);  // N1 is dead: masked M^-1
			oct_sync<DUO>();
			double L[N][N], dinv[N], y[N], x[N];
			UNR for (int i = 0; i < N; i++)
				UNR for (int j = 0; j <= i; j++) L[i][j] = sm.N1[i][j];
			oct_cholesky<N>(L, dinv);
			UNR for (int i = 0; i < N; i++) {
				double sy = sm.vec[4][i];
				UNR for (int k = 0; k < i; k++) sy = fma(-L[i][k], y[k], sy);
				y[i] = sy * dinv[i];
			}
			UNR for (int i = N - 1; i >= 0; i--) {
				double sx = y[i];
				UNR for (int k = i + 1; k < N; k++) sx = fma(-L[k][i], x[k], sx);
				x[i] = sx * dinv[i];
			}
			double yr = 0.0;
			UNR for (int i = 0; i < N; i++) yr = (i == rr) ? x[i] : yr;
			if (in_zone) tv = tj + tv - yr;
			if (P.torque_sat) {
				const double lim = md.effort[rr];
				tv = tv > lim ? lim : (tv < -lim ? -lim : tv);
			}
		}
	}
	if (P.gravity_comp) {
		const double* gv = md.gravity;
		const double ax_ = Ic[1] - Ic[0] * o[0], ay_ = Ic[2] - Ic[0] * o[1], az_ = Ic[3] - Ic[0] * o[2];  // m (c - o)
		const double cx = z[1] * az_ - z[2] * ay_, cy = z[2] * ax_ - z[0] * az_, cz = z[0] * ay_ - z[1] * ax_;
		const double tgr = -(cx * gv[0] + cy * gv[1] + cz * gv[2]);
		const double tgp = -Ic[0] * (z[0] * gv[0] + z[1] * gv[1] + z[2] * gv[2]);
		tv += rev ? tgr : tgp;
	}
	// a flagged instance holds its last valid torques (or gets them from the slow path launched behind) unless NaN was asked for
	// (with the slow tail on, the tail decides torques and status of a flagged instance: nothing is written for it here)
	const bool to_tail = GJ != 0 && P.slow_tail && singular;
	if (live && act && !singular) P.tau[(size_t)r * ld + b] = tv;
	if (live && act && singular && P.flag_nan && !to_tail) P.tau[(size_t)r * ld + b] = __builtin_nan("");
	if (live && r == 0 && !to_tail) P.status[b] = singular ? 1 : ((truncated ? 2 : 0) | (blended_i ? 8 : 0));
	if (live && r == 0 && singular && P.flag_list) {
		const int slot = atomicAdd(P.flag_count, 1);
		if (slot < P.B) P.flag_list[slot] = b;
	}  // work list of the device-side slow path
	if (P.flag_list && blockIdx.x == 0 && lane == 0) *P.flag_count_next = 0;                       // the other list: its last readers finished a cycle ago
	// integrators: commit what the control laws advanced them to unless the instance ended flagged (then nothing advances: the slow path
	// launched behind recomputes it from the untouched state, or it stays invalid).  Motion-force rows from the LDS block of the wavefront
	// that ran the law (B's block outlives B), joint rows from the lane that owns them.
	if (live && !singular) {
		if (track_mf && r < 6) P.integ2[0][(size_t)r * ld + b] = (ROLE == 1 ? smB : sm).ist[r];
		if (track_jt && (GJ == 1 ? r < mj : act)) P.integ2[1][(size_t)r * ld + b] = jt_ie_new;
	}
	STAMP(15);
	return live && to_tail;
}

// Slow tail (round 4): the instances of this wavefront that ended flagged -- outside SingularityHandler's non-singular branch with the blended
// strategies on (SingularityHandler.cpp:100-121, 146-158, 310-367), a reduced task with fewer than two directions left, an ambiguous rank
// gap -- are recomputed here, one after the other, by the general kernel's body (saip_wg_cycle.h: Jacobi eigen-pairs of the Gram matrix,
// U_ns / U_s split, blended type-1 / type-2 strategies with their per-instance state) on this wavefront's own LDS block, which the cycle no
// longer needs.  The instance's inputs are untouched (nothing was written for it, no integrator advanced), so the result is what the list
// launch behind the kernel used to produce -- without the launch, which cost 4.5 us per cycle whether or not anything was on the list.
// Everything the tail needs is in HBM or batch-uniform: no register of the ordinary path lives across it.
__device__ __forceinline__ void oct_slow_tail(const CycleParams& P, const bool flagged, void* lds) {
	const unsigned long long votes = __ballot(flagged);
	if (__builtin_expect(votes == 0ull, 1)) return;  // wave-uniform: the usual case
	static_assert(sizeof(WgSmem<8>) <= 8 * sizeof(OctInst), "the general kernel's block fits the eight instance blocks of a wavefront");
	WgSmem<8>& wsm = *reinterpret_cast<WgSmem<8>*>(lds);
	for (int g = 0; g < 8; g++) {
		const int l0 = OCT_ILV ? (((g >> 1) << 4) | (g & 1)) : (g << 3);  // the lane of joint 0 of instance g (octl_grp / octl_r)
		if (((votes >> l0) & 1ull) == 0ull) continue;                     // wave-uniform
		SAIP_WG_SYNC();  // the block's previous user (the cycle, or the instance before) is done with it
		wg_cycle<8, 64>(P, (int)blockIdx.x * 8 + g, wsm);
	}
}

// SIM (rollouts of stacks without internal OTGs and without a slow path behind): the wavefront that finished the cycle of its eight
// instances goes on to integrate them (saip_dynamics_oct_body.h) -- same lane layout, the torques it has just written, the LDS block
// the cycle no longer needs: one launch per closed-loop period.  A separate instantiation, so that the benchmarked kernel does not carry it.
template <bool FULL, int GJ, bool DUO = false, bool SIM = false, int NN = 7>
__global__ void __launch_bounds__(DUO ? 128 : 64) saip_cycle_oct(const CycleParams P) {
	static_assert(!SIM || NN == 7, "the fused integration step is the 7-dof one");
	__shared__ OctInst smem[DUO ? 16 : 8];
	static_assert(sizeof(OctInst) >= sizeof(OctDynInst), "the integration step reuses the cycle's LDS blocks");
	const bool wave_b = DUO && __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 1;
	__shared__ int wgflags[2];
	__shared__ double lawc[FULL ? 96 : 1];  // batch-uniform constants of the partial task's control law (the one wavefront of the workgroup that runs it)
#if !defined(SAIP_OCT_NO_BROWS)
	if (DUO && !FULL && GJ == 0) {
		if (threadIdx.x == 0) wgflags[0] = wgflags[1] = 0;
		__syncthreads();
	}
#endif
	bool flagged = false;
	if (DUO) {
		if (wave_b) oct_cycle_body<FULL, GJ, DUO ? 2 : 0, NN>(P, smem + 8, smem, smem + 8, wgflags, lawc);
		else flagged = oct_cycle_body<FULL, GJ, DUO ? 1 : 0, NN>(P, smem, smem, smem + 8, wgflags, lawc);
	} else {
		flagged = oct_cycle_body<FULL, GJ, 0, NN>(P, smem, smem, smem, nullptr, lawc);
	}
	// (the stacks whose blended branch lives in the kernel itself, GJ == 0, have nothing left for a tail; in the two-wavefront form wavefront A runs it:
	// B returned behind the first workgroup barrier)
	if (GJ != 0 && !SIM && !wave_b) oct_slow_tail(P, flagged, smem);
	if (SIM && !wave_b) {
		SimParams S;
		S.B = P.B;
		S.ld = P.ld;
		S.n = P.n;
		S.substeps = P.sim_substeps;
		S.dt = P.sim_dt;
		S.damping = P.sim_damping;
		S.gravity[0] = P.sim_gravity[0];
		S.gravity[1] = P.sim_gravity[1];
		S.gravity[2] = P.sim_gravity[2];
		S.model = P.model;
		S.q = const_cast<double*>(P.q);
		S.dq = const_cast<double*>(P.dq);
		S.tau = P.tau;
		S.ddq = nullptr;
		// (B's block in the two-wavefront form: B has left it for good at the last barrier; else this wavefront's own)
		integrate_oct_body(S, reinterpret_cast<OctDynInst*>(DUO ? smem + 8 : smem), (int)blockIdx.x, (int)(threadIdx.x & 63));
	}
}

// two wavefronts per instance group for the lean stack while the launch still fits the chip in one round (two workgroups per CU = one
// wavefront per SIMD: 4096 instances on 256 CUs); beyond that the second wavefront only competes for issue slots (measured, B = 6144:
// 14.3 us against 10.8 us).  SAIP_OCT_NO_DUO=1: the one-wavefront form, for A/B measurements.
static bool oct_duo_enabled(const int workgroups) {
	static const bool on = getenv("SAIP_OCT_NO_DUO") == nullptr;
	static int cus = 0;
	if (cus == 0) {
		int dev = 0, n = 0;
		if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
		cus = n;
	}
	return on && workgroups <= 2 * cus;
}

hipError_t launch_cycle_oct(const CycleParams& P, hipStream_t stream) {
	const dim3 grid((P.B + 7) / 8), block(64);
	const bool duo = oct_duo_enabled((int)grid.x);
	if (P.oct_general_joint == 2 && P.n == 8) {
		if (duo) hipLaunchKernelGGL((saip_cycle_oct<true, 2, true, false, 8>), grid, dim3(128), 0, stream, P);
		else hipLaunchKernelGGL((saip_cycle_oct<true, 2, false, false, 8>), grid, block, 0, stream, P);
	} else if (P.oct_general_joint == 2 && P.n == 6) {
		if (duo) hipLaunchKernelGGL((saip_cycle_oct<true, 2, true, false, 6>), grid, dim3(128), 0, stream, P);
		else hipLaunchKernelGGL((saip_cycle_oct<true, 2, false, false, 6>), grid, block, 0, stream, P);
	} else if (P.oct_general_joint == 2) {
		if (duo) hipLaunchKernelGGL((saip_cycle_oct<true, 2, true>), grid, dim3(128), 0, stream, P);
		else hipLaunchKernelGGL((saip_cycle_oct<true, 2>), grid, block, 0, stream, P);
	} else if (P.oct_general_joint == 1) {
		if (duo) hipLaunchKernelGGL((saip_cycle_oct<true, 1, true>), grid, dim3(128), 0, stream, P);
		else hipLaunchKernelGGL((saip_cycle_oct<true, 1>), grid, block, 0, stream, P);
	} else if (P.jla || P.lane_general || P.oct_partial_mf) {
		if (P.sim_substeps > 0) {
			if (duo) hipLaunchKernelGGL((saip_cycle_oct<true, 0, true, true>), grid, dim3(128), 0, stream, P);
			else hipLaunchKernelGGL((saip_cycle_oct<true, 0, false, true>), grid, block, 0, stream, P);
		} else if (duo) hipLaunchKernelGGL((saip_cycle_oct<true, 0, true>), grid, dim3(128), 0, stream, P);
		else hipLaunchKernelGGL((saip_cycle_oct<true, 0>), grid, block, 0, stream, P);
	} else {
		if (P.sim_substeps > 0) {
			if (duo) hipLaunchKernelGGL((saip_cycle_oct<false, 0, true, true>), grid, dim3(128), 0, stream, P);
			else hipLaunchKernelGGL((saip_cycle_oct<false, 0, false, true>), grid, block, 0, stream, P);
		} else if (duo) hipLaunchKernelGGL((saip_cycle_oct<false, 0, true>), grid, dim3(128), 0, stream, P);
		else hipLaunchKernelGGL((saip_cycle_oct<false, 0>), grid, block, 0, stream, P);
	}
	return hipGetLastError();
}

#if defined(SAIP_STAMP_OCT)
extern "C" int saip_debug_stamps_oct(unsigned long long* out, int n) {
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(saip_stamps_oct), sizeof(unsigned long long) * (n < 1024 * 64 ? n : 1024 * 64));
}
#endif
}  // namespace saip
