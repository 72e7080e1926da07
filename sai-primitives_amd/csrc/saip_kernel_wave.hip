// Control-cycle kernel for chains of 9..32 dof (BASELINE config 5: 30-dof chain, MotionForce + MotionForce + posture JointTask):
// ONE WAVEFRONT PER ROBOT INSTANCE, every matrix of the instance in that wavefront's registers in the operand layout of
// v_mfma_f64_16x16x4_f64 (saip_wave_prims.h), two wavefronts per SIMD (<= 256 registers per lane, 16 KB of LDS per instance), no
// s_barrier anywhere: the control flow of an instance is wave-uniform, so every data-dependent decision is a scalar branch.
//
// What it computes is what saip_cycle_wg<32,512> computes (reference call stack in saip_kernel_wg.hip:7-14):
//   SaiModel::updateModel()                       -> log-step prefix-product kinematics, composite inertias, M(q) straight into MFMA operand
//                                                    form, M^-1 by the blocked symmetric sweep on the matrix cores
//   RobotController::updateControllerTaskModels() -> /root/reference/src/RobotController.cpp:68-77; per task
//       MotionForceTask.cpp:247-268 + SingularityHandler.cpp:75-228 (non-singular branch): Jh = B^T Jw N_prec, the singular-value
//           certificate of the lane kernel on the Gram matrix (two MFMA squarings + LDL^T pivots of the shifted matrix),
//           Lambda = (Jh M^-1 Jh^T)^-1, Lambda_mod by decoupling type, N^T = I - Jh^T (Lambda Jh M^-1), N_prec <- N N_prec
//       JointTask.cpp:218-283: Jp = S N_prec, range basis by column-pivoted Gram-Schmidt with a clean-gap acceptance rule (rolled loop,
//           pivot column broadcast inside the 16-lane rows), U^T Jp through an LDS image that hands back both operand forms
//   RobotController::computeControlTorques()      -> RobotController.cpp:79-118, MotionForceTask.cpp:270-509 (saip_law.h on one lane),
//       SingularityHandler.cpp:297-309, JointTask.cpp:285-356; torque saturation, joint limit avoidance wrap, gravity compensation.
// Bounded inertia estimates (SingularityHandler.cpp:175-193): with c <= 8 clamped diagonal entries of M (index matrix E, n x c)
//   J M_BIE^-1 J^T = A - (T1 E) K^-1 (T1 E)^T,  K = C^-1 + E^T M^-1 E,  T1 = J M^-1   (Woodbury on the task side: no second n x n inverse).
//
// Everything outside the certified fast path -- a task outside SingularityHandler's non-singular branch, an ambiguous rank gap in a
// range basis, more than 8 clamped inertias, tasks with different BIE thresholds, a factorisation that meets a non-positive pivot --
// leaves the instance untouched (shadow integrators, no handler state) with status 1 on the device-side work list; the general kernel
// launched behind (saip_cycle_wg_list) recomputes exactly those instances.  On BASELINE config 5's distribution none are.
#include <hip/hip_runtime.h>
#include <math.h>

#include <type_traits>

#include "saip_device.h"
#include "saip_fk.h"
#include "saip_law.h"
#include "saip_wave_prims.h"

#if defined(SAIP_STAMP_WAVE)
// diagnostic build only (never shipped): s_memtime of lane 0 of the first 256 instances at the phase boundaries (tools/wave_stamps.py)
__device__ unsigned long long saip_stamps_wave[256][64];
#define WSTAMP(id)                                                                              \
	do {                                                                                        \
		if (lane == 0 && b < 256) saip_stamps_wave[b][id] = __builtin_amdgcn_s_memtime();       \
	} while (0)
#else
#define WSTAMP(id)
#endif

namespace saip {

struct WaveSmem {
	static constexpr int LDI = 33;
	double img[32 * LDI];  // staging for the kinematics scans, then the image of U^T Jp (joint task), read back in both operand forms
	double jd[32][12];     // per joint, about the world origin: motion axis S = (w, v), F = Ic S = (f, nO)
	double fr[32][12];     // world frame of every body: R (row-major), o
	double q[32], dq[32], tau[32], va[32], vb[32], vc[32], vd[32], act[32], diagM[32];
	double tg[32];         // jointGravityVector entry of every joint (gravity compensation): parked here from the inertia phase to the epilogue
	double goal[96];
	double vw[8], Fum[8], Ff[8], xp[4], Rc[12];
	double f8[SAIP_MAXT][8], ff8[SAIP_MAXT][8], g8[16];  // per motion-force task: B^T F_um, B^T F_f (range(P) coordinates); g = Lambda_mod fhat + ffhat
	int idx[16];
	int flag2;
};

// sin / cos of a joint angle: saip_fk.h's Cody-Waite + fdlibm kernels without the library fallback for |x| >= 1e5 (whose private arrays
// would put scratch into this kernel): such an angle flags the instance instead (the general kernel evaluates it)
__device__ __forceinline__ void wv_sincos(const double x, double* sn, double* cs) {
	const double k = rint(x * 6.36619772367581382433e-01);
	double r = fma(-k, 1.57079632673412561417e+00, x);
	r = fma(-k, 6.07710050650619224932e-11, r);
	const double z = r * r;
	const double ps = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03);
	const double s0 = fma(z * r, fma(z, ps, -1.66666666666666324348e-01), r);
	const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
	const double c0 = 1.0 - fma(0.5, z, -z * pc);
	const int q = (int)k & 3;
	const double sa = (q & 1) ? c0 : s0, ca = (q & 1) ? s0 : c0;
	*sn = (q & 2) ? -sa : sa;
	*cs = ((q + 1) & 2) ? -ca : ca;
}

__device__ __forceinline__ double wv_dot6(const double* S, const double* F) {  // S = (w, v), F = (f, nO): w . nO + v . f
	return fma(S[5], F[2], fma(S[4], F[1], fma(S[3], F[0], fma(S[2], F[5], fma(S[1], F[4], S[0] * F[3])))));
}

#ifndef SAIP_WAVE_OCC
#define SAIP_WAVE_OCC 2  // wavefronts per SIMD the register allocation aims at (A/B knob of tools/oct_variant.sh)
#endif
// JLA: RobotController::enableJointLimitAvoidance.  A separate instantiation because the wrap needs M^-1 once more AFTER the task loop: without it
// the stored M^-1 (32 registers) is dead as soon as the last task has used it.
// GENLAW: some motion-force task runs the general control law (velocity saturation, force / moment spaces, closed-loop force control);
// GENJT: some joint task needs a range basis (anything but ONE full joint task closing the hierarchy).  Both are batch-uniform properties of the
// stack, decided on the host (launch_cycle_wave).  They are template parameters because the register allocator sees every path of the task loop
// at once: with the general law (~100 live values on one lane) and the Gram-Schmidt path (six 32 x 32 forms live) compiled in, the stack of
// BASELINE config 5 -- which runs neither -- spilled its per-joint state at every turn of the law loop and M^-1 / N_prec at the head of the task
// loop (656 B of scratch per lane, 343 reloads per wavefront: 42 % of its cycles waiting for memory; round 4).
template <bool JLA, bool GENLAW, bool GENJT>
__global__ void __launch_bounds__(64, SAIP_WAVE_OCC) saip_cycle_wave(const CycleParams P) {
	__shared__ WaveSmem sm;
	const int b = saip_xcd_block(blockIdx.x, gridDim.x);  // (one instance per workgroup: the instances of an XCD are contiguous, see saip_oct_common.h)
	if (b >= P.B) return;
	const int lane = threadIdx.x, lo = lane & 15, hi = lane >> 4;
	const int j = lane & 31;  // joint owned in the per-joint phases (both halves of the wavefront do the same work)
	const int n = P.n, ld = P.ld;
	const ModelDev& md = *P.model;
	const bool jv = j < n;
	constexpr int LDI = WaveSmem::LDI;

	bool flagged = false;  // wave-uniform
	WSTAMP(0);
	// ---------------------------------------------------------------- state
	const double qj = jv ? P.q[(size_t)j * ld + b] : 0.0;
	const double dqj = jv ? P.dq[(size_t)j * ld + b] : 0.0;
	if (__ballot(jv && !(fabs(qj) < 1.0e5)) != 0ull) flagged = true;  // not a joint angle (or not finite): left to the general kernel
	if (lane < 32) {
		sm.q[j] = qj;
		sm.dq[j] = dqj;
		sm.tau[j] = 0.0;
	}
	if (lane == 0) sm.flag2 = 0;
	// ---------------------------------------------------------------- updateModel(): kinematics (joint j in lane j)
	const JointRec& jr = md.jrec[jv ? j : 0];
	const bool rev = jr.jtype == 1;
	double X[12];  // world frame of body j: R (row-major), o
	{
		const double ax = jr.axis[0], ay = jr.axis[1], az = jr.axis[2];
		double s, c;
		wv_sincos(qj, &s, &c);
		if (!rev) {
			s = 0.0;
			c = 1.0;
		}
		const double v = 1.0 - c;
		const double Rq[9] = {c + ax * ax * v,      ax * ay * v - az * s, ax * az * v + ay * s,
							  ay * ax * v + az * s, c + ay * ay * v,      ay * az * v - ax * s,
							  az * ax * v - ay * s, az * ay * v + ax * s, c + az * az * v};
		UNR for (int r = 0; r < 3; r++)
			UNR for (int cc = 0; cc < 3; cc++) X[3 * r + cc] = jr.R0[3 * r] * Rq[cc] + jr.R0[3 * r + 1] * Rq[3 + cc] + jr.R0[3 * r + 2] * Rq[6 + cc];
		const double a3[3] = {ax, ay, az};
		double d3[3];
		oct_mat3_vec(jr.R0, a3, d3);
		UNR for (int e = 0; e < 3; e++) X[9 + e] = jr.p0[e] + (rev ? 0.0 : d3[e] * qj);
		if (!jv) {
			UNR for (int e = 0; e < 12; e++) X[e] = (e == 0 || e == 4 || e == 8) ? 1.0 : 0.0;
		}
	}
	// log-step prefix product over the chain: T_j <- T_{j-d} o T_j, (Ra, oa) o (Rb, ob) = (Ra Rb, oa + Ra ob).  Joint j sits in lane j (and j + 32):
	// joints 0..15 and 16..31 are two 16-lane DPP rows, so the steps d = 1, 2, 4, 8 are row shifts (a VALU move per dword; lanes without a left
	// neighbour receive the identity) and only the last step crosses rows: joint 15's frame, read through the scalar unit, in front of every
	// joint of the second row.  (Round 4; the five steps used to be LDS round trips with two wavefront fences each.)
#define WV_FRAME_STEP(D)                                                                                                                  \
	if (D < n) {                                                                                                                          \
		double Xa[12], Xn[12];                                                                                                            \
		UNR for (int e = 0; e < 12; e++) Xa[e] = (e == 0 || e == 4 || e == 8) ? wv_shr<D, true>(X[e]) : wv_shr<D, false>(X[e]);           \
		UNR for (int r = 0; r < 3; r++) {                                                                                                 \
			UNR for (int cc = 0; cc < 3; cc++) Xn[3 * r + cc] = Xa[3 * r] * X[cc] + Xa[3 * r + 1] * X[3 + cc] + Xa[3 * r + 2] * X[6 + cc]; \
			Xn[9 + r] = Xa[9 + r] + Xa[3 * r] * X[9] + Xa[3 * r + 1] * X[10] + Xa[3 * r + 2] * X[11];                                     \
		}                                                                                                                                 \
		UNR for (int e = 0; e < 12; e++) X[e] = Xn[e];                                                                                    \
	}
	WV_FRAME_STEP(1)
	WV_FRAME_STEP(2)
	WV_FRAME_STEP(4)
	WV_FRAME_STEP(8)
#undef WV_FRAME_STEP
	if (n > 16) {
		double Xa[12], Xn[12];
		UNR for (int e = 0; e < 12; e++) Xa[e] = wv_readlane(X[e], 15);
		UNR for (int r = 0; r < 3; r++) {
			UNR for (int cc = 0; cc < 3; cc++) Xn[3 * r + cc] = Xa[3 * r] * X[cc] + Xa[3 * r + 1] * X[3 + cc] + Xa[3 * r + 2] * X[6 + cc];
			Xn[9 + r] = Xa[9 + r] + Xa[3 * r] * X[9] + Xa[3 * r + 1] * X[10] + Xa[3 * r + 2] * X[11];
		}
		UNR for (int e = 0; e < 12; e++) X[e] = (lane & 16) ? Xn[e] : X[e];
	}
	WSTAMP(1);
	// world axis, spatial inertia of body j about the world origin (m, h = m c, Io: xx yy zz xy xz yz)
	double z[3], Bd[10];
	{
		const double a3[3] = {jr.axis[0], jr.axis[1], jr.axis[2]}, cl[3] = {jr.com[0], jr.com[1], jr.com[2]};
		double c[3];
		oct_mat3_vec(X, a3, z);
		oct_mat3_vec(X, cl, c);
		UNR for (int e = 0; e < 3; e++) c[e] += X[9 + e];
		const double* I6 = jr.inertia;
		const double Il[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
		double RI[9], Iw[9];
		UNR for (int r = 0; r < 3; r++)
			UNR for (int cc = 0; cc < 3; cc++) RI[3 * r + cc] = X[3 * r] * Il[cc] + X[3 * r + 1] * Il[3 + cc] + X[3 * r + 2] * Il[6 + cc];
		UNR for (int r = 0; r < 3; r++)
			UNR for (int cc = 0; cc < 3; cc++) Iw[3 * r + cc] = RI[3 * r] * X[3 * cc] + RI[3 * r + 1] * X[3 * cc + 1] + RI[3 * r + 2] * X[3 * cc + 2];
		const double m = jr.mass, cc2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
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
		if (!jv) {
			UNR for (int e = 0; e < 10; e++) Bd[e] = 0.0;
		}
	}
	// composite inertias: log-step suffix sums (plain sums: everything is expressed about the world origin), the same way: row shifts, then the
	// second row's total (lane 16) on top of the first row
	UNR for (int e = 0; e < 10; e++) Bd[e] += wv_shl<1>(Bd[e]);
	UNR for (int e = 0; e < 10; e++) Bd[e] += wv_shl<2>(Bd[e]);
	UNR for (int e = 0; e < 10; e++) Bd[e] += wv_shl<4>(Bd[e]);
	UNR for (int e = 0; e < 10; e++) Bd[e] += wv_shl<8>(Bd[e]);
	if (n > 16) {
		UNR for (int e = 0; e < 10; e++) {
			const double up = wv_readlane(Bd[e], 16);
			Bd[e] += (lane & 16) ? 0.0 : up;
		}
	}
	// motion axis S_j = (w, v) and F_j = Ic_j S_j about the world origin
	double tg = 0.0;  // jointGravityVector entry of joint j
	{
		const double* o = X + 9;
		double w[3], v[3];
		if (rev) {
			w[0] = z[0]; w[1] = z[1]; w[2] = z[2];
			v[0] = o[1] * z[2] - o[2] * z[1];
			v[1] = o[2] * z[0] - o[0] * z[2];
			v[2] = o[0] * z[1] - o[1] * z[0];
		} else {
			w[0] = w[1] = w[2] = 0.0;
			v[0] = z[0]; v[1] = z[1]; v[2] = z[2];
		}
		const double m = Bd[0], hx = Bd[1], hy = Bd[2], hz = Bd[3];
		const double f[3] = {m * v[0] + (w[1] * hz - w[2] * hy), m * v[1] + (w[2] * hx - w[0] * hz), m * v[2] + (w[0] * hy - w[1] * hx)};
		const double nO[3] = {Bd[4] * w[0] + Bd[7] * w[1] + Bd[8] * w[2] + (hy * v[2] - hz * v[1]),
							  Bd[7] * w[0] + Bd[5] * w[1] + Bd[9] * w[2] + (hz * v[0] - hx * v[2]),
							  Bd[8] * w[0] + Bd[9] * w[1] + Bd[6] * w[2] + (hx * v[1] - hy * v[0])};
		if (lane < 32) {
			UNR for (int e = 0; e < 12; e++) sm.fr[j][e] = X[e];
			UNR for (int e = 0; e < 3; e++) {
				sm.jd[j][e] = jv ? w[e] : 0.0;
				sm.jd[j][3 + e] = jv ? v[e] : 0.0;
				sm.jd[j][6 + e] = jv ? f[e] : 0.0;
				sm.jd[j][9 + e] = jv ? nO[e] : 0.0;
			}
		}
		if (P.gravity_comp) {  // tau_g[j] = -S_j . (composite gravity wrench)
			const double* g = md.gravity;
			if (rev) {
				const double ax = hx - m * o[0], ay = hy - m * o[1], az = hz - m * o[2];  // m (c - o)
				const double cx = z[1] * az - z[2] * ay, cy = z[2] * ax - z[0] * az, cz = z[0] * ay - z[1] * ax;
				tg = -(cx * g[0] + cy * g[1] + cz * g[2]);
			} else {
				tg = -m * (z[0] * g[0] + z[1] * g[1] + z[2] * g[2]);
			}
			if (lane < 32) sm.tg[j] = tg;
		}
	}
	wv_sync();
	// ---------------------------------------------------------------- motion-force control laws (MotionForceTask.cpp:286-509), ahead of everything
	// that needs registers: a law depends on the kinematics only (control frame, vw = Jw dq, goals), runs on one lane and carries a
	// large set of temporaries (sigma projectors, ...) -- here nothing else is live yet.  F_um, F_f wait in LDS as range(P) coordinates.
	unsigned commit_lo = 0, commit_hi = 0;
	// goal rows of a motion-force task: 36, one per lane.  The NEXT task's are requested while this task's law runs (one live value); requesting every
	// task's rows at the start of the kernel was measured and lost (180.5 against 172.7 us per launch: eight live values across the kinematics).
	double gnext = 0.0;
	{
		int u = 0;
		while (u < P.nt && P.tasks[u].type != TASK_MOTION_FORCE) u++;
		if (u < P.nt && lane < P.tasks[u].goal_comps) gnext = saip_ldg(P.tasks[u].goal, (size_t)lane * ld + b);
	}
	for (int t = 0; t < P.nt; t++) {
		const TaskDev& tk = P.tasks[t];
		if (tk.type != TASK_MOTION_FORCE) continue;
		const bool track = tk.has_ki || P.integ_always;
		const int k = tk.k, bd = tk.body;
		{
			const double gcur = gnext;
			int u = t + 1;
			while (u < P.nt && P.tasks[u].type != TASK_MOTION_FORCE) u++;
			gnext = (u < P.nt && lane < P.tasks[u].goal_comps) ? saip_ldg(P.tasks[u].goal, (size_t)lane * ld + b) : 0.0;
			if (lane < tk.goal_comps) sm.goal[lane] = gcur;
		}
		double Rb[9], ob[3], pw[3], Rcf[9];
		UNR for (int e = 0; e < 9; e++) Rb[e] = sm.fr[bd][e];
		UNR for (int e = 0; e < 3; e++) ob[e] = sm.fr[bd][9 + e];
		oct_mat3_vec(Rb, tk.pos, pw);
		UNR for (int e = 0; e < 3; e++) pw[e] += ob[e];
		UNR for (int r = 0; r < 3; r++)
			UNR for (int cc = 0; cc < 3; cc++) Rcf[3 * r + cc] = Rb[3 * r] * tk.rot[cc] + Rb[3 * r + 1] * tk.rot[3 + cc] + Rb[3 * r + 2] * tk.rot[6 + cc];
		// task-frame velocities use the unprojected Jacobian (MotionForceTask.cpp:293-298): vw = Jw dq, column j = (w_j x p + v_j; w_j)
		double vwp[6] = {0, 0, 0, 0, 0, 0};
		UNR for (int ks = 0; ks < 8; ks++) {
			const int jj = 4 * ks + hi;
			const double* S = sm.jd[jj];
			const double dqv = (jj <= bd) ? sm.dq[jj] : 0.0;
			vwp[0] = fma(S[1] * pw[2] - S[2] * pw[1] + S[3], dqv, vwp[0]);
			vwp[1] = fma(S[2] * pw[0] - S[0] * pw[2] + S[4], dqv, vwp[1]);
			vwp[2] = fma(S[0] * pw[1] - S[1] * pw[0] + S[5], dqv, vwp[2]);
			vwp[3] = fma(S[0], dqv, vwp[3]);
			vwp[4] = fma(S[1], dqv, vwp[4]);
			vwp[5] = fma(S[2], dqv, vwp[5]);
		}
		UNR for (int e = 0; e < 6; e++) {
			const double sv = wv_hsum(vwp[e]);
			if (lane == 0) sm.vw[e] = sv;
		}
		if (lane == 0) {
			UNR for (int e = 0; e < 3; e++) sm.xp[e] = pw[e];
			UNR for (int e = 0; e < 9; e++) sm.Rc[e] = Rcf[e];
		}
		wv_sync();
		if (lane == 0) {
			double ip[3] = {0, 0, 0}, io[3] = {0, 0, 0}, ifm[6] = {0, 0, 0, 0, 0, 0};
			if (track)
				for (int i = 0; i < 3; i++) {
					ip[i] = saip_ldg(tk.integ, (size_t)i * ld + b);
					io[i] = saip_ldg(tk.integ, (size_t)(3 + i) * ld + b);
				}
			const bool closed = tk.cl_force || tk.cl_moment;
			if (closed)
				for (int i = 0; i < 6; i++) ifm[i] = saip_ldg(tk.integ, (size_t)(6 + i) * ld + b);
			bool ovf = false;
			if (GENLAW && tk.general_law) {
				// velocity saturation, force / moment spaces, open- or closed-loop force control
				law_motion_force_general<true>(&tk, sm.vw, sm.xp, sm.Rc, sm.goal, ip, io, ifm, sm.Fum, sm.Ff, tk.popc ? tk.popc + b : nullptr, ld, &ovf);
			} else {
				// the defaults (sigma_force = sigma_moment = 0, no velocity saturation): MotionForceTask.cpp:411-468 with the host's sigma matrices
				const double* G = sm.goal;
				const double* vw = sm.vw;
				double v[3], w[3], e3[3], tmp[3], oe[3] = {0, 0, 0}, dphi[3];
				if (tk.law_identity) {
					// full 6-dof task, no force / moment space (the reference default): the projections and sigma matrices are identities -- none of their
					// 36 products, none of the 36 scalar loads in front of them (round 4)
					UNR for (int i = 0; i < 3; i++) e3[i] = pw[i] - G[i];
					UNR for (int i = 0; i < 3; i++) ip[i] = fma(e3[i], tk.dt, ip[i]);
					UNR for (int i = 0; i < 3; i++) sm.Fum[i] = G[18 + i] - tk.kp_pos[i] * e3[i] - tk.kv_pos[i] * (vw[i] - G[12 + i]) - tk.ki_pos[i] * ip[i];
					UNR for (int c = 0; c < 3; c++) {  // orientationError(desired, current) = -1/2 sum_c Rc[:,c] x Rd[:,c]
						const double a0 = Rcf[c], a1 = Rcf[3 + c], a2 = Rcf[6 + c], b0 = G[3 + c], b1 = G[6 + c], b2 = G[9 + c];
						oe[0] -= 0.5 * (a1 * b2 - a2 * b1);
						oe[1] -= 0.5 * (a2 * b0 - a0 * b2);
						oe[2] -= 0.5 * (a0 * b1 - a1 * b0);
					}
					UNR for (int i = 0; i < 3; i++) io[i] = fma(oe[i], tk.dt, io[i]);
					UNR for (int i = 0; i < 3; i++) sm.Fum[3 + i] = G[21 + i] - tk.kp_ori[i] * oe[i] - tk.kv_ori[i] * (vw[3 + i] - G[15 + i]) - tk.ki_ori[i] * io[i];
					UNR for (int i = 0; i < 6; i++) sm.Ff[i] = 0.0;
				} else {
				UNR for (int i = 0; i < 3; i++) {
					v[i] = tk.Ppos[3 * i] * vw[0] + tk.Ppos[3 * i + 1] * vw[1] + tk.Ppos[3 * i + 2] * vw[2];
					w[i] = tk.Pori[3 * i] * vw[3] + tk.Pori[3 * i + 1] * vw[4] + tk.Pori[3 * i + 2] * vw[5];
					e3[i] = pw[i] - G[i];
				}
				UNR for (int i = 0; i < 3; i++) ip[i] += (tk.sig_p[3 * i] * e3[0] + tk.sig_p[3 * i + 1] * e3[1] + tk.sig_p[3 * i + 2] * e3[2]) * tk.dt;
				UNR for (int i = 0; i < 3; i++) tmp[i] = G[18 + i] - tk.kp_pos[i] * e3[i] - tk.kv_pos[i] * (v[i] - G[12 + i]) - tk.ki_pos[i] * ip[i];
				UNR for (int i = 0; i < 3; i++) sm.Fum[i] = tk.sig_p[3 * i] * tmp[0] + tk.sig_p[3 * i + 1] * tmp[1] + tk.sig_p[3 * i + 2] * tmp[2];
				UNR for (int c = 0; c < 3; c++) {  // orientationError(desired, current) = -1/2 sum_c Rc[:,c] x Rd[:,c]
					const double a0 = Rcf[c], a1 = Rcf[3 + c], a2 = Rcf[6 + c], b0 = G[3 + c], b1 = G[6 + c], b2 = G[9 + c];
					oe[0] -= 0.5 * (a1 * b2 - a2 * b1);
					oe[1] -= 0.5 * (a2 * b0 - a0 * b2);
					oe[2] -= 0.5 * (a0 * b1 - a1 * b0);
				}
				UNR for (int i = 0; i < 3; i++) dphi[i] = tk.sig_o[3 * i] * oe[0] + tk.sig_o[3 * i + 1] * oe[1] + tk.sig_o[3 * i + 2] * oe[2];
				UNR for (int i = 0; i < 3; i++) io[i] += dphi[i] * tk.dt;
				UNR for (int i = 0; i < 3; i++) tmp[i] = G[21 + i] - tk.kp_ori[i] * dphi[i] - tk.kv_ori[i] * (w[i] - G[15 + i]) - tk.ki_ori[i] * io[i];
				UNR for (int i = 0; i < 3; i++) sm.Fum[3 + i] = tk.sig_o[3 * i] * tmp[0] + tk.sig_o[3 * i + 1] * tmp[1] + tk.sig_o[3 * i + 2] * tmp[2];
				UNR for (int i = 0; i < 6; i++) sm.Ff[i] = 0.0;
				}
			}
			if (ovf) sm.flag2 = 1;
			if (closed)
				for (int i = 0; i < 6; i++) tk.integ_new[(size_t)(6 + i) * ld + b] = ifm[i];
			if (track)
				for (int i = 0; i < 3; i++) {
					tk.integ_new[(size_t)i * ld + b] = ip[i];
					tk.integ_new[(size_t)(3 + i) * ld + b] = io[i];
				}
		}
		if (track) commit_lo |= 1u << t;
		if (tk.cl_force || tk.cl_moment) commit_hi |= 1u << t;
		wv_sync();
		if (lane < 8) {  // fhat = Bm^T F_um, ffhat = Bm^T F_f
			double s1 = 0.0, s2 = 0.0;
			if (tk.law_identity) {  // Bm = I: no per-lane loads of its columns
				s1 = lane < 6 ? sm.Fum[lane < 6 ? lane : 0] : 0.0;
				s2 = lane < 6 ? sm.Ff[lane < 6 ? lane : 0] : 0.0;
			} else if (lane < k)
				for (int i = 0; i < 6; i++) {
					s1 = fma(tk.Bm[i * 6 + lane], sm.Fum[i], s1);
					s2 = fma(tk.Bm[i * 6 + lane], sm.Ff[i], s2);
				}
			sm.f8[t][lane] = s1;
			sm.ff8[t][lane] = s2;
		}
		wv_sync();
	}
	WSTAMP(2);
	// ---------------------------------------------------------------- M(q) in stored form: M_ij = S_min(i,j) . F_max(i,j); identity beyond n.
	// On the matrix cores (round 4): with S = (w, v) and F' = (nO, f) as 6 x n matrices, P = S^T F' holds S_i . F_j in its upper triangle and
	// P^T = F'^T S the same for the lower one -- two products of depth 6 (16 MFMAs, 8 LDS reads per lane) where every lane used to form 32 dot
	// products of six from 100 LDS reads.
	double Minv[8][2];
	{
		double Sst[2][2], Fst[2][2];  // stored forms: row = component 4 ks + hi (6, 7: padding), column = joint 16 t + lo
		UNR for (int ks = 0; ks < 2; ks++)
			UNR for (int t = 0; t < 2; t++) {
				const int c = 4 * ks + hi, cs = c < 6 ? c : 0;
				Sst[ks][t] = c < 6 ? sm.jd[16 * t + lo][cs] : 0.0;
				Fst[ks][t] = c < 6 ? sm.jd[16 * t + lo][cs < 3 ? 9 + cs : 3 + cs] : 0.0;  // F' = (nO, f): w . nO + v . f
			}
		double Pm[8][2], Pt[8][2];
		wv_mm<false, false>(Sst, Fst, Pm);  // S_i . F_j
		wv_mm<false, false>(Fst, Sst, Pt);  // S_j . F_i
		UNR for (int ks = 0; ks < 8; ks++) {
			const int i = 4 * ks + hi;
			UNR for (int t = 0; t < 2; t++) {
				const int jc = 16 * t + lo;
				double val = (i <= jc) ? Pm[ks][t] : Pt[ks][t];
				if (i >= n || jc >= n) val = (i == jc) ? 1.0 : 0.0;
				Minv[ks][t] = val;
				if (i == jc) sm.diagM[i] = val;
			}
		}
	}
	WSTAMP(3);
	{
		const double mp = wv_sweep<2>(Minv, (n + 3) >> 2, lo, hi);
		if (!(mp > 0.0)) flagged = true;  // (wv_sweep returns a scalar-unit value: these branches are scalar)
		wv_negate(Minv);
	}
	wv_sync();
	WSTAMP(4);
	// ---------------------------------------------------------------- bounded inertia estimates: clamped entries, K^-1
	double Kinv[2][1], Kmat[2][1];  // K^-1 and K itself (the posture task's Woodbury step wants K)
	int nclamp = 0, myidx = -1;  // column lo of the index matrix E (n x c) selects joint idx[lo]
	Kinv[0][0] = Kinv[1][0] = 0.0;
	Kmat[0][0] = Kmat[1][0] = 0.0;
	if (P.any_bie && !flagged) {
		const double thr = P.bie_thr;
		const bool cl = lane < 32 && jv && sm.diagM[j] < thr;
		const unsigned long long mask = __ballot(cl);
		const int pos = __popcll(mask & ((1ull << lane) - 1ull));
		nclamp = __popcll(mask);
		if (cl && pos < 8) sm.idx[pos] = j;
		wv_sync();
		if (nclamp > 8) {
			flagged = true;
		} else if (nclamp > 0) {
			myidx = lo < nclamp ? sm.idx[lo] : -1;
			// E^T M^-1 E (c x c) is the submatrix of M^-1 on the clamped joints: a gather through the LDS image (free here: N_prec moves in later) instead
			// of two products with the index matrix (24 MFMAs; round 4)
			double K4[4][1];
			UNR for (int ks = 0; ks < 8; ks++)
				UNR for (int tt = 0; tt < 2; tt++) sm.img[(4 * ks + hi) * LDI + 16 * tt + lo] = Minv[ks][tt];
			wv_sync();
			UNR for (int ks = 0; ks < 4; ks++) {
				const int i = 4 * ks + hi;
				const bool in = ks < 2 && i < nclamp && lo < nclamp;
				const int ri = in ? sm.idx[i < 8 ? i : 0] : 0;
				K4[ks][0] = in ? sm.img[ri * LDI + (myidx >= 0 ? myidx : 0)] : 0.0;
			}
			wv_sync();
			UNR for (int ks = 0; ks < 4; ks++) {
				const int i = 4 * ks + hi;
				if (i == lo) {
					if (i < nclamp) K4[ks][0] += 1.0 / (thr - sm.diagM[myidx]);
					else if (i < 8) K4[ks][0] = 1.0;
				}
			}
			Kmat[0][0] = K4[0][0];
			Kmat[1][0] = K4[1][0];
			const int nbk = (nclamp + 3) >> 2;  // c <= 4 clamped joints: one pivot block (the identity padding behind it is left alone)
			const double mp = wv_sweep<1>(K4, nbk, lo, hi);
			if (!(mp > 0.0)) flagged = true;
			Kinv[0][0] = -K4[0][0];
			Kinv[1][0] = nbk > 1 ? -K4[1][0] : K4[1][0];
		}
	}
	WSTAMP(5);
	// ---------------------------------------------------------------- task hierarchy
	// N_prec.  Stacks without a range-basis joint task (!GENJT) have no other use for the LDS image (it serves the Gram-Schmidt path): there N_prec
	// LIVES in the image between its uses -- read in stored form where a product wants it, in transposed form by the posture task, written back by
	// the rank-k update -- and its 32 registers are free in between (they used to spend the whole task loop in scratch memory: the register
	// allocator parked N_prec there at the loop head).  With the general joint task compiled in, N_prec stays in registers as before.
	constexpr bool NPL = !GENJT;
	double Np[8][2];
	auto np_load = [&]() {
		UNR for (int ks = 0; ks < 8; ks++)
			UNR for (int tt = 0; tt < 2; tt++) Np[ks][tt] = sm.img[(4 * ks + hi) * LDI + 16 * tt + lo];
	};
	auto np_store = [&]() {
		UNR for (int ks = 0; ks < 8; ks++)
			UNR for (int tt = 0; tt < 2; tt++) sm.img[(4 * ks + hi) * LDI + 16 * tt + lo] = Np[ks][tt];
	};
	if (!NPL) {
		UNR for (int ks = 0; ks < 8; ks++)
			UNR for (int t = 0; t < 2; t++) Np[ks][t] = (4 * ks + hi == 16 * t + lo && 4 * ks + hi < n) ? 1.0 : 0.0;
	}
	// N_prec is still the identity (wave-uniform): the first task's products with it are skipped ("first task skips the product with
	// N_prec = I").  Not in the GENJT instantiations: their one loop body would carry both versions of every product (measured in registers: 912
	// against 656 B of scratch per lane).
	bool np_ident = NPL;
	bool all_full = true;  // every task so far acts with its full rank (motion-force tasks: certified, else flagged; joint tasks: r == m)
	// stored Jb = Bm^T Jw (k x n) of a motion-force task: lane (lo, hi) owns task directions 4 ks + hi of joints 16 tt + lo
	auto jacobian_rows = [&](const TaskDev& tu, double (&Jb)[2][2]) {
		const int ku = tu.k, bu = tu.body;
		double Rb[9], ob[3], pw[3];
		UNR for (int e = 0; e < 9; e++) Rb[e] = sm.fr[bu][e];
		UNR for (int e = 0; e < 3; e++) ob[e] = sm.fr[bu][9 + e];
		oct_mat3_vec(Rb, tu.pos, pw);
		UNR for (int e = 0; e < 3; e++) pw[e] += ob[e];
		double bm[2][6];
		if (tu.law_identity) {  // Bm = I: row a of Jb is row a of the world Jacobian (unit vectors: no loads)
			UNR for (int ks = 0; ks < 2; ks++)
				UNR for (int i = 0; i < 6; i++) bm[ks][i] = (4 * ks + hi == i) ? 1.0 : 0.0;
		} else {
			UNR for (int ks = 0; ks < 2; ks++)
				UNR for (int i = 0; i < 6; i++) bm[ks][i] = (4 * ks + hi < ku) ? tu.Bm[i * 6 + (4 * ks + hi < 6 ? 4 * ks + hi : 0)] : 0.0;
		}
		UNR for (int tt = 0; tt < 2; tt++) {
			const int jj = 16 * tt + lo;
			const double* S = sm.jd[jj];
			const bool on = jj <= bu;
			const double w0 = on ? S[0] : 0.0, w1 = on ? S[1] : 0.0, w2 = on ? S[2] : 0.0;
			const double c0 = on ? (w1 * pw[2] - w2 * pw[1] + S[3]) : 0.0, c1 = on ? (w2 * pw[0] - w0 * pw[2] + S[4]) : 0.0,
						 c2 = on ? (w0 * pw[1] - w1 * pw[0] + S[5]) : 0.0;
			UNR for (int ks = 0; ks < 2; ks++)
				Jb[ks][tt] = fma(w2, bm[ks][5], fma(w1, bm[ks][4], fma(w0, bm[ks][3], fma(c2, bm[ks][2], fma(c1, bm[ks][1], c0 * bm[ks][0])))));
		}
	};
	// One task of the hierarchy.  MODE 0: either kind (the loop of the GENJT instantiations); 1: motion-force tasks only; 2: the joint task only.  The
	// stacks without a range-basis joint task (!GENJT) are "motion-force tasks, then at most ONE full joint task": they run a loop over MODE 1 and
	// then MODE 2 once, so that no turn of a loop carries the register needs of both kinds (the allocator spills what lives across a loop at its
	// head: with both kinds in one body that was M^-1 or N_prec).
	auto run_task = [&](const int t, auto mode_tag) __attribute__((always_inline)) {
		constexpr int MODE = decltype(mode_tag)::value;
		const TaskDev& tk = P.tasks[t];
		const bool need_N = t + 1 < P.nt;
		const bool track = tk.has_ki || P.integ_always;
		WSTAMP(8 + 10 * t);
		if (tk.type != TASK_MOTION_FORCE)
			for (int e = lane; e < tk.goal_comps; e += 64) sm.goal[e] = saip_ldg(tk.goal, (size_t)e * ld + b);
		if (tk.decoupling == DEC_BIE && tk.bie_threshold != P.bie_thr) {
			flagged = true;
			return;
		}
		if (MODE != 2 && tk.type == TASK_MOTION_FORCE) {
			const int k = tk.k, bd = tk.body;
			// control point of the task (uniform LDS address: a broadcast read)
			double Rb[9], ob[3], pw[3];
			UNR for (int e = 0; e < 9; e++) Rb[e] = sm.fr[bd][e];
			UNR for (int e = 0; e < 3; e++) ob[e] = sm.fr[bd][9 + e];
			oct_mat3_vec(Rb, tk.pos, pw);
			UNR for (int e = 0; e < 3; e++) pw[e] += ob[e];
			// Jb^T = Jw^T Bm (n x k) in stored form: lane (lo, hi) owns task direction lo of joints 4 ks + hi; column j of
			// JWorldFrame is (w_j x p + v_j; w_j) for the joints up to the body (MotionForceTask.cpp:262)
			double bmc[6];
			if (tk.law_identity) {
				UNR for (int i = 0; i < 6; i++) bmc[i] = (lo == i) ? 1.0 : 0.0;
			} else {
				UNR for (int i = 0; i < 6; i++) bmc[i] = (lo < k) ? tk.Bm[i * 6 + (lo < 6 ? lo : 0)] : 0.0;
			}
			double Jbt[8][1];
			UNR for (int ks = 0; ks < 8; ks++) {
				const int jj = 4 * ks + hi;
				const double* S = sm.jd[jj];
				const bool on = jj <= bd;
				const double w0 = on ? S[0] : 0.0, w1 = on ? S[1] : 0.0, w2 = on ? S[2] : 0.0;
				const double c0 = on ? (w1 * pw[2] - w2 * pw[1] + S[3]) : 0.0, c1 = on ? (w2 * pw[0] - w0 * pw[2] + S[4]) : 0.0,
							 c2 = on ? (w0 * pw[1] - w1 * pw[0] + S[5]) : 0.0;
				Jbt[ks][0] = fma(w2, bmc[5], fma(w1, bmc[4], fma(w0, bmc[3], fma(c2, bmc[2], fma(c1, bmc[1], c0 * bmc[0])))));
			}
			__builtin_amdgcn_sched_barrier(0);
			WSTAMP(9 + 10 * t);
			double Jh[2][2], Jht[8][1];
			if (np_ident) {  // N_prec = I: Jh = Jb, both forms straight from the joint data (32 MFMAs less)
				jacobian_rows(tk, Jh);
				UNR for (int ks = 0; ks < 8; ks++) Jht[ks][0] = Jbt[ks][0];
			} else {
				if (NPL) np_load();
				wv_mm<false, false>(Jbt, Np, Jh);   // Jh = Jb N_prec (k x n)
				wv_mm<false, false>(Np, Jbt, Jht);  // Jh^T (n x k)
			}
			WSTAMP(10 + 10 * t);
			// singular values of the projected Jacobian (SingularityHandler.cpp:78-121): certificate on G = Jh Jh^T instead of an SVD
			bool certified;
			{
				double G[2][1], G2[2][1], G4[2][1];
				wv_mm<false, false>(Jht, Jht, G);
				double trp = 0.0;
				UNR for (int ks = 0; ks < 2; ks++) trp += (lo == 4 * ks + hi) ? G[ks][0] : 0.0;
				const double tr = wv_wavesum(trp);
				const bool nonfinite = !(tr - tr == 0.0);
				const double itr = 1.0 / tr;
				UNR for (int ks = 0; ks < 2; ks++) G[ks][0] *= itr;
				wv_mm<false, false>(G, G, G2);
				wv_mm<false, false>(G2, G2, G4);
				const double t8 = wv_wavesum(fma(G4[0][0], G4[0][0], G4[1][0] * G4[1][0]));
				const double u = sqrt(sqrt(sqrt(t8)));
				const bool ok_s0 = (u * tk.cert_kroot * tr >= tk.s_abs_tol * tk.s_abs_tol);
				const double shift = tk.s_max * tk.s_max * u;
				double Gc[4][1];
				UNR for (int ks = 0; ks < 4; ks++) {
					const int i = 4 * ks + hi;
					double val = ks < 2 ? G[ks < 2 ? ks : 0][0] : 0.0;
					if (i == lo && i < k) val -= shift;
					if ((i >= k || lo >= k) && i < 8 && lo < 8) val = (i == lo) ? 1.0 : 0.0;
					Gc[ks][0] = val;
				}
				const double mp = wv_sweep<1, true>(Gc, 2, lo, hi);  // (the pivots are all the certificate wants)
				certified = wv_uniform((mp > 1e-13) && ok_s0 && (tr > 0.0) && !nonfinite);
			}
			// (no branch on the certificate yet: its chain -- three dependent products, a sweep -- and the Lambda chain below are independent; in one
			// basic block the scheduler interleaves their MFMAs instead of waiting out each one's result latency)
			WSTAMP(11 + 10 * t);
			double T1[2][2], T1t[8][1], A2[2][1], Lam[4][1], Lmod[4][1];
			wv_mm<false, false>(Jht, Minv, T1);   // Jh M^-1 (k x n)
			wv_mm<false, false>(Minv, Jht, T1t);  // M^-1 Jh^T (n x k)
			wv_mm<false, false>(T1t, Jht, A2);    // Jh M^-1 Jh^T
			auto pad_sweep = [&](const double (&Ain)[2][1], double (&Out)[4][1]) -> bool {
				UNR for (int ks = 0; ks < 4; ks++) {
					const int i = 4 * ks + hi;
					double val = ks < 2 ? Ain[ks < 2 ? ks : 0][0] : 0.0;
					if ((i >= k || lo >= k) && i < 8 && lo < 8) val = (i == lo) ? 1.0 : 0.0;
					Out[ks][0] = val;
				}
				const double mp = wv_sweep<1>(Out, 2, lo, hi);
				wv_negate(Out);
				return mp > 0.0;
			};
			const bool lam_ok = pad_sweep(A2, Lam);
			if (!certified || !lam_ok) {  // outside the non-singular branch, or too close to call: the general kernel decides (exact eigen-solve)
				flagged = true;
				return;
			}
			WSTAMP(12 + 10 * t);
			if (tk.decoupling == DEC_FULL) {
				UNR for (int ks = 0; ks < 4; ks++) Lmod[ks][0] = Lam[ks][0];
			} else if (tk.decoupling == DEC_IMPEDANCE) {
				UNR for (int ks = 0; ks < 4; ks++) Lmod[ks][0] = (4 * ks + hi == lo && lo < 8) ? 1.0 : 0.0;
			} else if (nclamp == 0) {
				UNR for (int ks = 0; ks < 4; ks++) Lmod[ks][0] = Lam[ks][0];
			} else {
				double Em[8][1], Ytt[2][1], Zk[2][1], AB[2][1];
				UNR for (int ks = 0; ks < 8; ks++) Em[ks][0] = (myidx == 4 * ks + hi) ? 1.0 : 0.0;
				wv_mm<false, false>(Em, T1t, Ytt);    // (T1 E)^T  (c x k)
				wv_mm<false, false>(Kinv, Ytt, Zk);   // K^-1 (T1 E)^T
				AB[0][0] = A2[0][0];
				AB[1][0] = A2[1][0];
				wv_mm<true, true>(Ytt, Zk, AB);       // A - (T1 E) K^-1 (T1 E)^T
				if (!pad_sweep(AB, Lmod)) {
					flagged = true;
					return;
				}
			}
			WSTAMP(13 + 10 * t);
			// g = Lambda_mod B^T F_um + B^T F_f (SingularityHandler.cpp:307-309); the control law itself ran ahead of the task algebra
			wv_sync();
			{
				double Lm2[2][1] = {{Lmod[0][0]}, {Lmod[1][0]}}, fr[2], y[1];
				wv_load_rows(sm.f8[t], hi, fr);
				wv_matvec(Lm2, fr, y);
				if (lane < 16) sm.g8[lane] = (lane < 8) ? y[0] + sm.ff8[t][lane] : 0.0;
			}
			wv_sync();
			{
				double gr[2], y[2];
				wv_load_rows(sm.g8, hi, gr);
				wv_matvec(Jh, gr, y);  // tau += Jh^T g
				if (hi == 0) {
					sm.tau[lo] += y[0];
					sm.tau[16 + lo] += y[1];
				}
			}
			// computeTorques(tau_prec): the reference's _Lambda is never written => no compensation
			WSTAMP(14 + 10 * t);
			if (need_N) {  // N_prec <- N N_prec = N_prec - Jbar (Jh N_prec) with Jbar^T = Lambda T1: a rank-k update in place (24 MFMAs, no n x n temporaries)
				double Lm2[2][1] = {{Lam[0][0]}, {Lam[1][0]}}, Jbart[2][2], JhN[2][2];
				wv_mm<false, false>(Lm2, T1, Jbart);
				if (np_ident) {  // Jh N_prec = Jh, N_prec = I
					UNR for (int ks = 0; ks < 2; ks++)
						UNR for (int tt = 0; tt < 2; tt++) JhN[ks][tt] = Jh[ks][tt];
					UNR for (int ks = 0; ks < 8; ks++)
						UNR for (int tt = 0; tt < 2; tt++) Np[ks][tt] = (4 * ks + hi == 16 * tt + lo && 4 * ks + hi < n) ? 1.0 : 0.0;
				} else {
					if (NPL) np_load();
					wv_mm<false, false>(Jht, Np, JhN);
				}
				wv_mm<true, true>(Jbart, JhN, Np);
				if (NPL) np_store();
				np_ident = false;
			}
			wv_sync();
			WSTAMP(15 + 10 * t);
		} else if (MODE != 1) {
			// ------------------------------------------------------------ JointTask
			const int m = tk.m;
			wv_sync();
			// control law per task row (JointTask.cpp:299-345): f -> vb, then xa = ddq_d - S M^-1 tau_prec (JointTask.cpp:287-291)
			if (track) commit_lo |= 1u << t;
			{
				double tr8[8], y[2];
				wv_load_rows(sm.tau, hi, tr8);
				wv_matvec(Minv, tr8, y);  // w = M^-1 tau_prec
				if (hi == 0) {
					sm.vd[lo] = y[0];
					sm.vd[16 + lo] = y[1];
				}
			}
			wv_sync();
			if (lane < 32) {
				double fv = 0.0, xa = 0.0;
				if (j < m) {
					double cur = 0.0, vel = 0.0, sw = 0.0;
					if (tk.s_identity) {
						cur = sm.q[j];  // (from LDS, not the registers of the kernel's first lines: those would live across the whole task loop)
						vel = sm.dq[j];
						sw = sm.vd[j];
					} else {
						for (int jj = 0; jj < n; jj++) {
							const double sv = tk.S[j * n + jj];
							cur = fma(sv, sm.q[jj], cur);
							vel = fma(sv, sm.dq[jj], vel);
							sw = fma(sv, sm.vd[jj], sw);
						}
					}
					const double e = cur - sm.goal[j];
					double ie = track ? saip_ldg(tk.integ, (size_t)j * ld + b) : 0.0;
					ie += e * tk.dt;  // JointTask.cpp:323-324
					if (track) tk.integ_new[(size_t)j * ld + b] = ie;
					if (tk.vel_sat) {  // JointTask.cpp:327-341
						double vdes = -tk.kp[j] * tk.kvinv[j] * e - tk.ki[j] * tk.kvinv[j] * ie;
						vdes = fmin(fmax(vdes, -tk.sat[j]), tk.sat[j]);
						fv = -tk.kv[j] * (vel - vdes);
					} else {
						fv = -tk.kp[j] * e - tk.kv[j] * (vel - sm.goal[m + j]) - tk.ki[j] * ie;  // :342-345
					}
					xa = sm.goal[2 * m + j] - sw;
				}
				sm.vb[j] = fv;
				sm.vc[j] = xa;
				sm.va[j] = 0.0;   // a1 = U^T xa, filled by the Gram-Schmidt loop
				sm.act[j] = 0.0;  // b1 = U^T f
			}
			if (tk.s_identity && m == n && all_full && !need_N) {
				// ---- full joint task (always the last of a hierarchy: RobotController.cpp:33-41) WITHOUT a range basis.  Jp = N_prec and
				// U Lambda U^T = X^+ with X = N_prec M^-1 N_prec^T, so tau = N_prec^T (X^+ a + X_B^+ f).  range(X) = range(N_prec) is the
				// common null space of the raw task Jacobians C above (full rank each): for any gamma > 0
				//     (X + gamma C^T C)^-1 = X^+ + (C^T C)^+ / gamma     and     N_prec^T (C^T C)^+ = 0        (C N_prec = 0),
				// hence tau = N_prec^T ((X + gamma C^T C)^-1 a + (X_B + gamma C^T C)^-1 f): two SPD sweeps on the matrix cores instead of a pivoted
				// Gram-Schmidt (n - sum k sequential steps) and the two reduced inverses.  matrixRangeBasis' rank decision (sigma_i / sigma_0 >= 1e-3,
				// JointTask.cpp:233) is certified instead of computed: N_prec is a projector, so its non-zero singular values are >= 1 and
				// sigma_0 <= ||N_prec||_F; ||N_prec||_F <= 100 keeps every one of them a factor 10 inside the tolerance (else: general kernel).
				const bool same = tk.decoupling == DEC_FULL || (tk.decoupling == DEC_BIE && nclamp == 0), ident = tk.decoupling == DEC_IMPEDANCE;
				if (ident) {  // Lambda_mod = I acts in range coordinates (U U^T f): impedance-type posture tasks keep the Gram-Schmidt path of the general kernel
					flagged = true;
					return;
				}
				double Xm[8][2], Zst[8][1];
				{
					double NpT[8][2], Rm[8][2];  // stored N_prec^T through the LDS image, then M^-1 N_prec^T
					if (!NPL) {
						np_store();
						wv_sync();
					} else if (np_ident) {  // (a full joint task alone: nothing has written the image yet)
						UNR for (int ks = 0; ks < 8; ks++)
							UNR for (int tt = 0; tt < 2; tt++) sm.img[(4 * ks + hi) * LDI + 16 * tt + lo] = (4 * ks + hi == 16 * tt + lo && 4 * ks + hi < n) ? 1.0 : 0.0;
						wv_sync();
					}
					UNR for (int ks = 0; ks < 8; ks++)
						UNR for (int tt = 0; tt < 2; tt++) NpT[ks][tt] = sm.img[(16 * tt + lo) * LDI + 4 * ks + hi];
					double fs = 0.0;
					UNR for (int ks = 0; ks < 8; ks++)
						UNR for (int tt = 0; tt < 2; tt++) fs = fma(NpT[ks][tt], NpT[ks][tt], fs);
					const double fro2 = wv_uniform(wv_wavesum(fs));  // ||N_prec||_F^2: the rank certificate above
					if (!(fro2 <= 1.0e4)) {
						flagged = true;
						return;
					}
					wv_mm<false, false>(Minv, NpT, Rm);
					wv_mm<false, false>(NpT, Rm, Xm);  // X = N_prec M^-1 N_prec^T
					if (!same) {
						double Em[8][1];
						UNR for (int ks = 0; ks < 8; ks++) Em[ks][0] = (myidx == 4 * ks + hi) ? 1.0 : 0.0;
						wv_mm<false, false>(Rm, Em, Zst);  // Z = N_prec M^-1 E (n x c)
					}
				}
				__builtin_amdgcn_sched_barrier(0);
				// Y = X + gamma C^T C, C = the raw Jacobians of the tasks above, gamma = trace X / trace C^T C (any positive value would do: it only
				// balances the two parts).  C^T C is never formed: trace(Jb^T Jb) = |Jb|_F^2 gives gamma first, then every task adds its
				// (sqrt(gamma) Jb)^T (sqrt(gamma) Jb) straight into X on the matrix cores.
				double dx = 0.0, dc = 0.0;
				UNR for (int ks = 0; ks < 8; ks++)
					UNR for (int tt = 0; tt < 2; tt++)
						if (4 * ks + hi == 16 * tt + lo) dx += Xm[ks][tt];
				for (int u = 0; u < t; u++) {
					const TaskDev& tu = P.tasks[u];
					if (tu.type == TASK_MOTION_FORCE) {
						double Jb[2][2];
						jacobian_rows(tu, Jb);
						UNR for (int ks = 0; ks < 2; ks++)
							UNR for (int tt = 0; tt < 2; tt++) dc = fma(Jb[ks][tt], Jb[ks][tt], dc);
					} else {
						UNR for (int ks = 0; ks < 8; ks++)
							UNR for (int tt = 0; tt < 2; tt++) {
								const double sv = (4 * ks + hi < tu.m && 16 * tt + lo < n) ? tu.S[(4 * ks + hi) * n + 16 * tt + lo] : 0.0;
								dc = fma(sv, sv, dc);
							}
					}
				}
				const double trX = wv_uniform(wv_wavesum(dx)), trC = wv_uniform(wv_wavesum(dc));
				const double gam = trC > 0.0 ? trX / trC : 0.0, sg = sqrt(gam);
				for (int u = 0; u < t; u++) {
					const TaskDev& tu = P.tasks[u];
					if (tu.type == TASK_MOTION_FORCE) {
						double Jb[2][2];
						jacobian_rows(tu, Jb);
						UNR for (int ks = 0; ks < 2; ks++)
							UNR for (int tt = 0; tt < 2; tt++) Jb[ks][tt] *= sg;
						wv_mm<true, false>(Jb, Jb, Xm);  // += gamma Jb^T Jb
					} else {
						double Ss[8][2];  // stored S (m x n), full row rank under its N_prec (all_full)
						UNR for (int ks = 0; ks < 8; ks++)
							UNR for (int tt = 0; tt < 2; tt++)
								Ss[ks][tt] = (4 * ks + hi < tu.m && 16 * tt + lo < n) ? sg * tu.S[(4 * ks + hi) * n + 16 * tt + lo] : 0.0;
						wv_mm<true, false>(Ss, Ss, Xm);  // += gamma S^T S
					}
				}
				__builtin_amdgcn_sched_barrier(0);
				const int nbn = (n + 3) >> 2;
				double zr[8], y2[2], zsum[2] = {0.0, 0.0};
				wv_pad_identity(Xm, n, 4 * nbn, lo, hi);
				{
					const double mp = wv_sweep<2>(Xm, nbn, lo, hi);  // Xm = -Y^-1
					if (!(mp > 0.0)) flagged = true;
				}
				if (flagged) return;
				// z = Y^-1 a + (X_B + gamma C^T C)^-1 f with X_B = X - Z K^-1 Z^T (bounded inertia estimates on the clamped joints).  The second inverse is
				// a Woodbury step on the first (round 4: it used to be a second 32 x 32 sweep -- 48 MFMAs, eight pivot blocks and 32 more live registers):
				//     (Y - Z K^-1 Z^T)^-1 = Y^-1 + Y^-1 Z (K - Z^T Y^-1 Z)^-1 Z^T Y^-1      =>      z = Y^-1 (a + f + Z w),   w = (K - Z^T Y^-1 Z)^-1 Z^T Y^-1 f,
				// one c x c sweep (c <= 8 clamped joints); K - Z^T Y^-1 Z is the Schur complement of [Y Z; Z^T K], positive definite exactly when X_B + gamma C^T C is.
				wv_load_rows(sm.vc, hi, zr);  // a
				{
					double fr8[8];
					wv_load_rows(sm.vb, hi, fr8);  // f
					if (!same) {
						double Wst[8][1], ZtW[2][1], Sm4[4][1], yu[2], ur[8], zu[1], zur[2], wc[1];
						wv_mm<false, true>(Xm, Zst, Wst);    // Y^-1 Z (n x c): Xm = -Y^-1
						wv_mm<false, false>(Zst, Wst, ZtW);  // Z^T Y^-1 Z (c x c)
						UNR for (int ks = 0; ks < 4; ks++) {
							const int i = 4 * ks + hi;
							double val = ks < 2 ? Kmat[ks < 2 ? ks : 0][0] - ZtW[ks < 2 ? ks : 0][0] : 0.0;
							if ((i >= nclamp || lo >= nclamp) && i < 8 && lo < 8) val = (i == lo) ? 1.0 : 0.0;
							Sm4[ks][0] = val;
						}
						const double mp = wv_sweep<1>(Sm4, (nclamp + 3) >> 2, lo, hi);  // -> -(K - Z^T Y^-1 Z)^-1 (one pivot block for c <= 4)
						if (!(mp > 0.0)) flagged = true;
						wv_matvec(Xm, fr8, yu);  // -Y^-1 f, column distribution
						wv_sync();
						if (hi == 0) {
							sm.vd[lo] = -yu[0];
							sm.vd[16 + lo] = -yu[1];
						}
						wv_sync();
						wv_load_rows(sm.vd, hi, ur);
						wv_matvec(Zst, ur, zu);  // Z^T Y^-1 f (c entries)
						if (lane < 16) sm.g8[lane] = zu[0];
						wv_sync();
						wv_load_rows(sm.g8, hi, zur);
						double Sm2[2][1] = {{Sm4[0][0]}, {Sm4[1][0]}};
						wv_matvec(Sm2, zur, wc);  // -w, column distribution: entry lo in every row of lanes
						// Z w in ROW distribution straight from the stored form: (Z w)[4 ks + hi] = sum over lo of Z[4 ks + hi][lo] w[lo]
						UNR for (int ks = 0; ks < 8; ks++) fr8[ks] -= wv_rowsum(Zst[ks][0] * wc[0]);
					}
					UNR for (int ks = 0; ks < 8; ks++) zr[ks] += fr8[ks];
				}
				if (flagged) return;
				wv_matvec(Xm, zr, y2);  // (the sweep leaves -inverse)
				zsum[0] -= y2[0];
				zsum[1] -= y2[1];
				wv_sync();
				if (hi == 0) {
					sm.vd[lo] = zsum[0];
					sm.vd[16 + lo] = zsum[1];
				}
				wv_sync();
				wv_load_rows(sm.vd, hi, zr);
				{
					// N_prec once more from its LDS image (written above for the transpose, untouched since): its 32 registers are free for the two
					// sweeps in between instead of living in scratch memory
					double NpL[8][2];
					UNR for (int ks = 0; ks < 8; ks++)
						UNR for (int tt = 0; tt < 2; tt++) NpL[ks][tt] = sm.img[(4 * ks + hi) * LDI + 16 * tt + lo];
					wv_matvec(NpL, zr, y2);  // tau += N_prec^T z
				}
				if (hi == 0) {
					sm.tau[lo] += y2[0];
					sm.tau[16 + lo] += y2[1];
				}
				wv_sync();
				WSTAMP(13 + 10 * t);
				return;  // (!need_N: this was the last task -- leaving the loop here tells the register allocator that M^-1 and N_prec are dead)
			}
			if (!GENJT) {  // (not reached: the host gives stacks with such a joint task the GENJT instantiation)
				flagged = true;
				return;
			}
			// Jp = S N_prec (JointTask.cpp:231)
			double Wc[8][2];
			if (tk.s_identity) {
				UNR for (int ks = 0; ks < 8; ks++)
					UNR for (int tt = 0; tt < 2; tt++) Wc[ks][tt] = Np[ks][tt];
			} else {
				double St[8][2];  // stored S^T: S[16 tt + lo][4 ks + hi]
				UNR for (int ks = 0; ks < 8; ks++)
					UNR for (int tt = 0; tt < 2; tt++) {
						const int row = 16 * tt + lo, col = 4 * ks + hi;
						St[ks][tt] = (row < m && col < n) ? tk.S[row * n + col] : 0.0;
					}
				wv_mm<false, false>(St, Np, Wc);
			}
			wv_sync();
			WSTAMP(9 + 10 * t);
			// matrixRangeBasis(Jp) (JointTask.cpp:233): column-pivoted Gram-Schmidt, accepted only on a clean rank gap (next pivot below
			// 1e-10 or above 1e-2 of the first); row s of U^T Jp goes to the LDS image, U^T xa / U^T f are taken on the fly
			const int rb = P.rank_bound[t] < m ? P.rank_bound[t] : m;
			bool need_exact = false, going = true;
			int rank = 0;
			double nr[2], c0 = 1.0;
			UNR for (int tt = 0; tt < 2; tt++) {
				double s = 0.0;
				UNR for (int ks = 0; ks < 8; ks++) s = fma(Wc[ks][tt], Wc[ks][tt], s);
				nr[tt] = wv_hsum(s);
			}
			for (int s = 0; s <= rb; s++) {
				double best = nr[0];
				int jb = lo;
				if (nr[1] > nr[0]) {
					best = nr[1];
					jb = 16 + lo;
				}
				wv_row_argmax(best, jb);
				best = wv_uniform(best);
				if (s == 0) {
					const double frob2 = wv_uniform(wv_rowsum(nr[0] + nr[1]));
					c0 = best;
					if (frob2 < 1e-6) going = false;  // ||Jp||_F < 1e-3: empty range
					else if (best < 1e-5 || rb == 0) {
						going = false;
						need_exact = true;
					}
				} else if (going) {
					const double ratio = best * oct_rcp(c0);
					if (ratio < 1e-20) going = false;
					else if (ratio < 1e-4 || s == rb) {
						going = false;
						need_exact = true;
					}
				}
				if (!going || s == rb) break;
				const int jbu = __builtin_amdgcn_readfirstlane(jb);
				const double inv = oct_rsqrt(best);
				const int src = (jbu & 15) + 16 * hi;
				double u[8];
				UNR for (int ks = 0; ks < 8; ks++) u[ks] = __shfl((jbu & 16) ? Wc[ks][1] : Wc[ks][0], src) * inv;  // U[:, s] in row distribution
				double d[2], pa = 0.0, pb = 0.0;
				UNR for (int tt = 0; tt < 2; tt++) {
					double acc = 0.0;
					UNR for (int ks = 0; ks < 8; ks++) acc = fma(u[ks], Wc[ks][tt], acc);
					d[tt] = wv_hsum(acc);
				}
				UNR for (int ks = 0; ks < 8; ks++) {
					pa = fma(u[ks], sm.vc[4 * ks + hi], pa);
					pb = fma(u[ks], sm.vb[4 * ks + hi], pb);
				}
				pa = wv_hsum(pa);
				pb = wv_hsum(pb);
				if (hi == 0) {
					sm.img[s * LDI + lo] = d[0];
					sm.img[s * LDI + 16 + lo] = d[1];
				}
				if (lane == 0) {
					sm.va[s] = pa;
					sm.act[s] = pb;
				}
				UNR for (int tt = 0; tt < 2; tt++) {
					double acc = 0.0;
					UNR for (int ks = 0; ks < 8; ks++) {
						const double v = fma(-u[ks], d[tt], Wc[ks][tt]);
						Wc[ks][tt] = v;
						acc = fma(v, v, acc);
					}
					nr[tt] = wv_hsum(acc);
				}
				rank = s + 1;
			}
			if (need_exact) {  // ambiguous rank gap: the general kernel decides with the reference's tolerances (eigen-solve of the Gram matrix)
				flagged = true;
				return;
			}
			const int r = rank;
			if (r != m) all_full = false;
			wv_sync();
			WSTAMP(10 + 10 * t);
			if (r > 0) {
				// both operand forms of U^T Jp come from the LDS image whenever they are needed: no more than five 32 x 32 matrices are
				// ever live (M^-1, N_prec and three of this block's)
				auto load_Jh = [&](double (&Jh)[8][2]) {
					UNR for (int ks = 0; ks < 8; ks++)
						UNR for (int tt = 0; tt < 2; tt++) Jh[ks][tt] = (4 * ks + hi) < r ? sm.img[(4 * ks + hi) * LDI + 16 * tt + lo] : 0.0;  // U^T Jp (r x n)
				};
				auto load_Jht = [&](double (&Jht)[8][2]) {
					UNR for (int ks = 0; ks < 8; ks++)
						UNR for (int tt = 0; tt < 2; tt++) Jht[ks][tt] = (16 * tt + lo) < r ? sm.img[(16 * tt + lo) * LDI + 4 * ks + hi] : 0.0;  // its transpose (n x r)
				};
				const int nb = (r + 3) >> 2;
				const bool same = tk.decoupling == DEC_FULL || (tk.decoupling == DEC_BIE && nclamp == 0), ident = tk.decoupling == DEC_IMPEDANCE;
				double A[8][2], Ytt[2][2];
				{
					double Jht[8][2], T1t[8][2];
					load_Jht(Jht);
					wv_mm<false, false>(Minv, Jht, T1t);  // M^-1 Jh^T (n x r)
					wv_mm<false, false>(T1t, Jht, A);     // Jh M^-1 Jh^T
					if (!same && !ident) {
						double Em[8][1];
						UNR for (int ks = 0; ks < 8; ks++) Em[ks][0] = (myidx == 4 * ks + hi) ? 1.0 : 0.0;
						wv_mm<false, false>(Em, T1t, Ytt);  // (T1 E)^T (c x r)
					}
				}
				__builtin_amdgcn_sched_barrier(0);
				WSTAMP(11 + 10 * t);
				// Lambda_mod (JointTask.cpp:253-267) first: its action on b1 = U^T f, then Lambda itself in place
				double ymod[2];
				{
					double b1r[8];
					wv_load_rows(sm.act, hi, b1r);
					if (ident) {
						ymod[0] = sm.act[lo];
						ymod[1] = sm.act[16 + lo];
					} else if (!same) {
						double LM[8][2], Zk[2][2];
						UNR for (int ks = 0; ks < 8; ks++)
							UNR for (int tt = 0; tt < 2; tt++) LM[ks][tt] = A[ks][tt];
						wv_mm<false, false>(Kinv, Ytt, Zk);  // K^-1 (T1 E)^T
						wv_mm<true, true>(Ytt, Zk, LM);      // A - (T1 E) K^-1 (T1 E)^T
						wv_pad_identity(LM, r, 4 * nb, lo, hi);
						const double mp = wv_sweep<2>(LM, nb, lo, hi);
						if (!(mp > 0.0)) flagged = true;
						wv_negate(LM);
						wv_matvec(LM, b1r, ymod);
					}
					__builtin_amdgcn_sched_barrier(0);
					wv_pad_identity(A, r, 4 * nb, lo, hi);
					const double mp = wv_sweep<2>(A, nb, lo, hi);
					if (!(mp > 0.0)) flagged = true;
					wv_negate(A);  // Lambda
					if (same) wv_matvec(A, b1r, ymod);
				}
				if (flagged) return;
				WSTAMP(12 + 10 * t);
				{
					double a1r[8], y[2];
					wv_load_rows(sm.va, hi, a1r);
					wv_matvec(A, a1r, y);  // g = Lambda a1 + Lambda_mod b1 (JointTask.cpp:348-351, 287-291)
					if (hi == 0) {
						sm.vd[lo] = (lo < r) ? y[0] + ymod[0] : 0.0;
						sm.vd[16 + lo] = (16 + lo < r) ? y[1] + ymod[1] : 0.0;
					}
				}
				wv_sync();
				if (!need_N) {
					double Jh[8][2], gr[8], y[2];
					load_Jh(Jh);
					wv_load_rows(sm.vd, hi, gr);
					wv_matvec(Jh, gr, y);  // tau += Jh^T g
					if (hi == 0) {
						sm.tau[lo] += y[0];
						sm.tau[16 + lo] += y[1];
					}
				}
				WSTAMP(13 + 10 * t);
				if (need_N) {
					double Jbart[8][2];
					{
						double Jht[8][2], T1[8][2];
						load_Jht(Jht);
						wv_mm<false, false>(Jht, Minv, T1);  // Jh M^-1 (r x n)
						wv_mm<false, false>(A, T1, Jbart);   // Lambda T1
					}
					__builtin_amdgcn_sched_barrier(0);
					double Jh[8][2], gr[8], y[2];
					load_Jh(Jh);
					wv_load_rows(sm.vd, hi, gr);
					wv_matvec(Jh, gr, y);  // tau += Jh^T g
					if (hi == 0) {
						sm.tau[lo] += y[0];
						sm.tau[16 + lo] += y[1];
					}
					UNR for (int ks = 0; ks < 8; ks++)
						UNR for (int tt = 0; tt < 2; tt++)
							if (4 * ks + hi >= r) Jbart[ks][tt] = 0.0;  // the identity padding of Lambda is not part of it
					__builtin_amdgcn_sched_barrier(0);
					// N_prec <- N N_prec = N_prec - Jbar (Jh N_prec), in place
					double Jht[8][2], JhN[8][2];
					load_Jht(Jht);
					wv_mm<false, false>(Jht, Np, JhN);
					wv_mm<true, true>(Jbart, JhN, Np);
					np_ident = false;
				}
			}
			// r == 0: no controllable direction left, N = I (JointTask.cpp:234-239)
			wv_sync();
		}
	};
	if (GENJT) {
		for (int t = 0; t < P.nt && !flagged; t++) run_task(t, std::integral_constant<int, 0>());
	} else {
		int t = 0;
		for (; t < P.nt && !flagged && P.tasks[t].type == TASK_MOTION_FORCE; t++) run_task(t, std::integral_constant<int, 1>());
		if (t < P.nt && !flagged) run_task(t, std::integral_constant<int, 2>());  // (the host gives this instantiation only to stacks whose one joint task is the last task)
	}
	wv_sync();
	WSTAMP(60);
	// ---------------------------------------------------------------- post-processing, RobotController.cpp:86-116
	double tv = sm.tau[j];
	if (P.torque_sat && jv) {
		const double lim = md.effort[j];
		tv = tv > lim ? lim : (tv < -lim ? -lim : tv);
	}
	if (JLA && P.jla && !flagged) {
		// tau = JLA.computeTorques(tau) + N_constraints^T tau,  N_constraints^T = I - S^T (S M^-1 S^T)^-1 S M^-1 (padded to n x n)
		bool actj = false;
		double jl = 0.0;
		if (jv) jl = jla_joint(sm.q[j], sm.dq[j], md.q_lower[j], md.q_upper[j], md.vel_limit[j], md.effort[j], tv, &actj);
		const unsigned long long am = __ballot(actj && lane < 32);
		wv_sync();
		if (lane < 32) {
			sm.act[j] = actj ? 1.0 : 0.0;
			sm.vc[j] = tv;
		}
		wv_sync();
		if (am != 0ull) {
			double tr8[8], y[2], ar[8];
			wv_load_rows(sm.vc, hi, tr8);
			wv_load_rows(sm.act, hi, ar);
			wv_matvec(Minv, tr8, y);  // M^-1 tau
			const double ac[2] = {sm.act[lo], sm.act[16 + lo]};
			if (hi == 0) {
				sm.vd[lo] = ac[0] != 0.0 ? y[0] : 0.0;
				sm.vd[16 + lo] = ac[1] != 0.0 ? y[1] : 0.0;
			}
			double Wm[8][2];
			UNR for (int ks = 0; ks < 8; ks++)
				UNR for (int tt = 0; tt < 2; tt++)
					Wm[ks][tt] = (ar[ks] != 0.0 && ac[tt] != 0.0) ? Minv[ks][tt] : ((4 * ks + hi == 16 * tt + lo) ? 1.0 : 0.0);
			const double mp = wv_sweep<2>(Wm, (n + 3) >> 2, lo, hi);
			if (!(mp > 0.0)) flagged = true;
			wv_sync();
			double v2r[8];
			wv_load_rows(sm.vd, hi, v2r);
			wv_matvec(Wm, v2r, y);  // -(S M^-1 S^T)^-1 S M^-1 tau
			const double yy = (lane & 16) ? y[1] : y[0];
			if (actj) tv = jl + tv + yy;
			if (P.torque_sat && jv) {
				const double lim = md.effort[j];
				tv = tv > lim ? lim : (tv < -lim ? -lim : tv);
			}
		}
	}
	if (P.gravity_comp) tv += sm.tg[j];
	// a flagged instance holds its last valid torques (or gets them from the general kernel launched behind) unless NaN was asked for
	if (lane < 32 && jv) {
		if (!flagged) P.tau[(size_t)j * ld + b] = tv;
		else if (P.flag_nan) P.tau[(size_t)j * ld + b] = __builtin_nan("");
	}
	if (lane == 0) {
		P.status[b] = flagged ? 1 : (sm.flag2 ? 4 : 0);
		if (flagged && P.flag_list) {
			const int slot = atomicAdd(P.flag_count, 1);
			if (slot < P.B) P.flag_list[slot] = b;
		}
	}
	WSTAMP(61);
	if (P.flag_list && b == 0 && lane == 0) *P.flag_count_next = 0;  // the other list: its last readers finished a cycle ago
	if (!flagged) {  // integrators: commit what the control laws advanced them to (a flagged instance advances nothing)
		for (int t = 0; t < P.nt; t++) {
			const TaskDev& tk = P.tasks[t];
			const int rows = tk.type == TASK_MOTION_FORCE ? 6 : tk.m;
			if ((commit_lo >> t) & 1u)
				for (int i = lane; i < rows; i += 64) tk.integ[(size_t)i * ld + b] = saip_ldg(tk.integ_new, (size_t)i * ld + b);
			if ((commit_hi >> t) & 1u)
				for (int i = lane; i < 6; i += 64) tk.integ[(size_t)(6 + i) * ld + b] = saip_ldg(tk.integ_new, (size_t)(6 + i) * ld + b);
		}
	}
}

#if defined(SAIP_STAMP_WAVE)
extern "C" int saip_debug_stamps_wave(unsigned long long* out) {
	return hipMemcpyFromSymbol(out, HIP_SYMBOL(saip_stamps_wave), sizeof(unsigned long long) * 256 * 64) == hipSuccess ? 0 : -1;
}
#endif
template <bool JLA, bool GENLAW>
static void launch_wave_jt(const bool genjt, const CycleParams& P, hipStream_t stream) {
	if (genjt) hipLaunchKernelGGL((saip_cycle_wave<JLA, GENLAW, true>), dim3(P.B), dim3(64), 0, stream, P);
	else hipLaunchKernelGGL((saip_cycle_wave<JLA, GENLAW, false>), dim3(P.B), dim3(64), 0, stream, P);
}
hipError_t launch_cycle_wave(const CycleParams& P, hipStream_t stream) {
	// P.lane_general: some task runs a general control law; P.wave_general_joint: some joint task needs a range basis (saip_engine.cpp)
	const bool genlaw = P.lane_general != 0, genjt = P.wave_general_joint != 0;
	if (P.jla) {
		if (genlaw) launch_wave_jt<true, true>(genjt, P, stream);
		else launch_wave_jt<true, false>(genjt, P, stream);
	} else {
		if (genlaw) launch_wave_jt<false, true>(genjt, P, stream);
		else launch_wave_jt<false, false>(genjt, P, stream);
	}
	return hipGetLastError();
}

}  // namespace saip
