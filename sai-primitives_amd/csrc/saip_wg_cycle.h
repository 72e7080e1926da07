// One control cycle of ONE robot instance by a workgroup of T threads, every matrix of the instance resident in LDS: the body of the general
// kernel (saip_kernel_wg.hip: one workgroup per instance, or a fixed grid striding over the list of flagged instances), and -- with T = 64 and a
// wavefront-scope SAIP_WG_SYNC -- the in-kernel slow tail of the eight-lane kernels (saip_kernel_oct.hip, saip_kernel_octjf.hip), which
// recompute the instances they could not finish on their own, by now dead, LDS block instead of handing them to a second launch.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>

#include "saip_device.h"
#include "saip_law.h"
#include "saip_wg_linalg.h"
#include "saip_fk.h"

#if defined(SAIP_STAMP_WG)
// diagnostic build only (never shipped): s_memtime log of block 0 / thread 0 after every barrier of the kernel body (time, source line)
__device__ unsigned long long saip_stamps_wg[4096];
__device__ int saip_stamps_wg_pos;
#undef SYNC
#define SYNC()                                                                 \
	do {                                                                       \
		__syncthreads();                                                       \
		if (blockIdx.x == 0 && threadIdx.x == 0) {                             \
			const int p_ = saip_stamps_wg_pos++;                               \
			if (p_ < 2047) {                                                   \
				saip_stamps_wg[2 * p_] = __builtin_amdgcn_s_memtime();         \
				saip_stamps_wg[2 * p_ + 1] = __LINE__;                         \
			}                                                                  \
		}                                                                      \
	} while (0)
#define MARK()                                                                 \
	do {                                                                       \
		if (blockIdx.x == 0 && threadIdx.x == 0) {                             \
			const int p_ = saip_stamps_wg_pos++;                               \
			if (p_ < 2047) {                                                   \
				saip_stamps_wg[2 * p_] = __builtin_amdgcn_s_memtime();         \
				saip_stamps_wg[2 * p_ + 1] = 100000 + __LINE__;                \
			}                                                                  \
		}                                                                      \
	} while (0)
#else
#define MARK()
#endif

namespace saip {

template <int NMAX>
struct WgSmem {
	static constexpr int LD = NMAX + 1;
	static constexpr int MS = NMAX * LD;
	static constexpr int NWORK = 9;
	double Minv[MS], MBinv[MS], NpA[MS], NpB[MS];
	double Nx[MS];  // per-task entry points: N_ns of a blended task (getTaskNullspace = N_js N_ns)
	double W[NWORK][MS];
	double fkA[NMAX * 12], fkB[NMAX * 12];  // per body: R (9, row-major) + o (3), world frame
	double zax[NMAX * 3];                   // world joint axes
	double body[NMAX * 10], comp[NMAX * 10];  // spatial inertia about the world origin: m, h(3)=m*c, Io(6: xx yy zz xy xz yz)
	double Jw[6 * LD];
	double q[NMAX], dq[NMAX], tau[NMAX], v0[NMAX], v1[NMAX], v2[NMAX], v3[NMAX], v4[NMAX], dv[NMAX];
	double goal[3 * NMAX > 36 ? 3 * NMAX : 36];
	double vw[6], Fum[6], Ff[6], xp[3], Rc[9];
	double cs[4 * (NMAX / 2 + 1)];
	int colidx[NMAX];
	int flag, rank;
	int flag2;  // passivity-observer window overflow (status bit 2)
	double bieK[64];  // bie_inverse_lowrank: the c x c Woodbury matrix and the clamped indices (+ their count)
	int bieIdx[9];
};

// one control cycle (or one task's model / torque half, CycleParams::single_task) of instance b by the calling workgroup
template <int NMAX, int T>
__device__ __forceinline__ void wg_cycle(const CycleParams& P, const int b, WgSmem<NMAX>& sm) {
	using SM = WgSmem<NMAX>;
	constexpr int LD = SM::LD;
	const int tid = T == 64 ? (int)(threadIdx.x & 63u) : (int)threadIdx.x;  // (T = 64: also the one wavefront of a larger workgroup that runs an eight-lane kernel's slow tail)
	const int n = P.n;
	const ModelDev& md = *P.model;
	// e / n for the element loops (0 <= e < 2048, n <= 32) without the ~40-instruction integer division of the hardware-less path:
	// multiply by ceil(2^18 / n) and shift (exact while e (n ceil(2^18 / n) - 2^18) < 2^18, i.e. for e < 4096)
	const unsigned divn_m = ((1u << 18) + (unsigned)n - 1u) / (unsigned)n;
	auto divn = [divn_m](const int e) { return (int)(((unsigned)e * divn_m) >> 18); };

	// ---------------------------------------------------------------- state
	if (tid == 0) sm.flag2 = 0;
	if (tid < n) {
		sm.q[tid] = P.q[(size_t)tid * P.ld + b];
		sm.dq[tid] = P.dq[(size_t)tid * P.ld + b];
		sm.tau[tid] = P.ext_tau_prec ? P.ext_tau_prec[(size_t)tid * P.ld + b] : 0.0;  // computeTorques(tau_prec), TemplateTask.h:59
	}
	SYNC();
	// ---------------------------------------------------------------- updateModel(): kinematics
	// local transforms X_j = (R0_j * Rot(axis_j, q_j), p0_j [+ R0_j axis_j q_j])
	for (int j = tid; j < n; j += T) {
		double* X = sm.fkA + 12 * j;
		const double* R0 = md.R0[j];
		const double ax = md.axis[j][0], ay = md.axis[j][1], az = md.axis[j][2];
		if (md.jtype[j] == 1) {
			double s, c;
			sincos(sm.q[j], &s, &c);
			double v = 1.0 - c;
			double Rq[9] = {c + ax * ax * v,      ax * ay * v - az * s, ax * az * v + ay * s,
							ay * ax * v + az * s, c + ay * ay * v,      ay * az * v - ax * s,
							az * ax * v - ay * s, az * ay * v + ax * s, c + az * az * v};
			for (int r = 0; r < 3; r++)
				for (int cc = 0; cc < 3; cc++) X[3 * r + cc] = R0[3 * r] * Rq[cc] + R0[3 * r + 1] * Rq[3 + cc] + R0[3 * r + 2] * Rq[6 + cc];
			X[9] = md.p0[j][0];
			X[10] = md.p0[j][1];
			X[11] = md.p0[j][2];
		} else {
			for (int e = 0; e < 9; e++) X[e] = R0[e];
			double a[3] = {ax, ay, az}, d[3];
			mat3_mul_vec(R0, a, d);
			for (int e = 0; e < 3; e++) X[9 + e] = md.p0[j][e] + d[e] * sm.q[j];
		}
	}
	SYNC();
	// log-step prefix product: T_j <- T_{j-d} o T_j,  (Ra,oa) o (Rb,ob) = (Ra Rb, oa + Ra ob)
	double* fk = sm.fkA;
	double* fk2 = sm.fkB;
	for (int d = 1; d < n; d <<= 1) {
		for (int e = tid; e < n * 12; e += T) {
			int j = e / 12, c = e - 12 * j;
			const double* Xb = fk + 12 * j;
			double val;
			if (j < d) {
				val = Xb[c];
			} else {
				const double* Xa = fk + 12 * (j - d);
				if (c < 9) {
					int r = c / 3, cc = c - 3 * r;
					val = Xa[3 * r] * Xb[cc] + Xa[3 * r + 1] * Xb[3 + cc] + Xa[3 * r + 2] * Xb[6 + cc];
				} else {
					int r = c - 9;
					val = Xa[9 + r] + Xa[3 * r] * Xb[9] + Xa[3 * r + 1] * Xb[10] + Xa[3 * r + 2] * Xb[11];
				}
			}
			fk2[e] = val;
		}
		SYNC();
		double* t_ = fk;
		fk = fk2;
		fk2 = t_;
	}
	// world axes + per-body spatial inertia about the world origin
	for (int j = tid; j < n; j += T) {
		const double* R = fk + 12 * j;
		const double* o = R + 9;
		double a[3] = {md.axis[j][0], md.axis[j][1], md.axis[j][2]}, z[3], c[3], cl[3] = {md.com[j][0], md.com[j][1], md.com[j][2]};
		mat3_mul_vec(R, a, z);
		mat3_mul_vec(R, cl, c);
		for (int e = 0; e < 3; e++) {
			sm.zax[3 * j + e] = z[e];
			c[e] += o[e];
		}
		const double* I6 = md.inertia[j];
		double Il[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]}, RI[9], Iw[9];
		for (int r = 0; r < 3; r++)
			for (int cc = 0; cc < 3; cc++) RI[3 * r + cc] = R[3 * r] * Il[cc] + R[3 * r + 1] * Il[3 + cc] + R[3 * r + 2] * Il[6 + cc];
		for (int r = 0; r < 3; r++)
			for (int cc = 0; cc < 3; cc++) Iw[3 * r + cc] = RI[3 * r] * R[3 * cc] + RI[3 * r + 1] * R[3 * cc + 1] + RI[3 * r + 2] * R[3 * cc + 2];
		double m = md.mass[j], cc2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
		double* Bd = sm.body + 10 * j;
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
	}
	SYNC();
	// composite inertias: suffix sums (plain sums, everything is expressed about the world origin)
	for (int e = tid; e < n * 10; e += T) {
		int j = e / 10, c = e - 10 * j;
		double s = 0.0;
		for (int l = j; l < n; l++) s += sm.body[10 * l + c];
		sm.comp[e] = s;
	}
	SYNC();
	// ---------------------------------------------------------------- M(q): M_ij = S_i . (Ic_j S_j), i <= j
	double* Mm = sm.W[0];
	double* MB = sm.W[1];
	for (int e = tid; e < n * n; e += T) {
		int i = divn(e), j = e - i * n;
		if (i > j) continue;
		const double* Ic = sm.comp + 10 * j;
		const double* zj = sm.zax + 3 * j;
		const double* oj = fk + 12 * j + 9;
		double w[3], v[3];  // spatial motion of joint j about the world origin: angular w, linear v
		if (md.jtype[j] == 1) {
			w[0] = zj[0]; w[1] = zj[1]; w[2] = zj[2];
			v[0] = oj[1] * zj[2] - oj[2] * zj[1];
			v[1] = oj[2] * zj[0] - oj[0] * zj[2];
			v[2] = oj[0] * zj[1] - oj[1] * zj[0];
		} else {
			w[0] = w[1] = w[2] = 0.0;
			v[0] = zj[0]; v[1] = zj[1]; v[2] = zj[2];
		}
		const double m = Ic[0], hx = Ic[1], hy = Ic[2], hz = Ic[3];
		double f[3] = {m * v[0] + (w[1] * hz - w[2] * hy), m * v[1] + (w[2] * hx - w[0] * hz), m * v[2] + (w[0] * hy - w[1] * hx)};
		double nO[3] = {Ic[4] * w[0] + Ic[7] * w[1] + Ic[8] * w[2] + (hy * v[2] - hz * v[1]),
						Ic[7] * w[0] + Ic[5] * w[1] + Ic[9] * w[2] + (hz * v[0] - hx * v[2]),
						Ic[8] * w[0] + Ic[9] * w[1] + Ic[6] * w[2] + (hx * v[1] - hy * v[0])};
		const double* zi = sm.zax + 3 * i;
		const double* oi = fk + 12 * i + 9;
		double val;
		if (md.jtype[i] == 1) {
			double vi[3] = {oi[1] * zi[2] - oi[2] * zi[1], oi[2] * zi[0] - oi[0] * zi[2], oi[0] * zi[1] - oi[1] * zi[0]};
			val = zi[0] * nO[0] + zi[1] * nO[1] + zi[2] * nO[2] + vi[0] * f[0] + vi[1] * f[1] + vi[2] * f[2];
		} else {
			val = zi[0] * f[0] + zi[1] * f[1] + zi[2] * f[2];
		}
		Mm[i * LD + j] = val;
		Mm[j * LD + i] = val;
	}
	SYNC();
	// does any task use bounded inertia estimates?  (M_BIE: diagonal clamped from below, SingularityHandler.cpp:176-182)
	bool any_bie = false;
	double bie_thr = 0.0;
	for (int t = 0; t < P.nt; t++)
		if (P.tasks[t].decoupling == DEC_BIE) {
			any_bie = true;
			bie_thr = P.tasks[t].bie_threshold;  // per-task thresholds that differ are handled below by recomputing
		}
	spd_inverse<T>(Mm, sm.Minv, sm.W[2], sm.W[3], sm.dv, n, LD, tid);
	double cur_bie_thr = -1.0;
	if (any_bie) {
		if (!bie_inverse_lowrank<T>(Mm, sm.Minv, sm.MBinv, sm.bieK, sm.bieIdx, sm.W[2], bie_thr, n, LD, tid)) {
			for (int e = tid; e < n * n; e += T) {
				int i = divn(e), j = e - i * n;
				double v = Mm[i * LD + j];
				if (i == j && v < bie_thr) v = bie_thr;
				MB[i * LD + j] = v;
			}
			SYNC();
			spd_inverse<T>(MB, sm.MBinv, sm.W[2], sm.W[3], sm.dv, n, LD, tid);
		}
		cur_bie_thr = bie_thr;
	}
	// keep M for per-task BIE thresholds that differ from cur_bie_thr: W[8] is reserved for it
	double* Mkeep = sm.W[8];
	for (int e = tid; e < n * n; e += T) {
		int i = divn(e), j = e - i * n;
		Mkeep[i * LD + j] = Mm[i * LD + j];
	}
	// N_prec = I
	double* Np = sm.NpA;
	double* Np2 = sm.NpB;
	for (int e = tid; e < n * n; e += T) {
		int i = divn(e), j = e - i * n;
		Np[i * LD + j] = P.ext_nprec ? P.ext_nprec[(size_t)e * P.ld + b] : ((i == j) ? 1.0 : 0.0);  // updateTaskModel(N_prec), TemplateTask.h:43
	}
	SYNC();
	// per-task entry points (TemplateTask.h:43-60): one task, split into its model and its torque half
	const int t_first = P.single_task >= 0 ? P.single_task : 0;
	const int t_last = P.single_task >= 0 ? P.single_task + 1 : P.nt;
	const bool do_law = P.phase != 1;    // updateTaskModel computes no torques (and advances neither integrators nor OTG)
	const bool do_state = P.phase != 2 && !P.diag;  // computeTorques does not classify (SingularityHandler::classifySingularity runs in updateTaskModel, .cpp:227)
	auto dump_nullspaces = [&](const double* Ntask, const double* Ntotal) {  // [n*n][ld] row-major per instance
		for (int e = tid; e < n * n; e += T) {
			int i = divn(e), j = e - i * n;
			if (P.out_N) P.out_N[(size_t)e * P.ld + b] = Ntask ? Ntask[i * LD + j] : ((i == j) ? 1.0 : 0.0);
			if (P.out_Ntot) P.out_Ntot[(size_t)e * P.ld + b] = Ntotal[i * LD + j];
		}
	};

	unsigned commit_lo = 0, commit_hi = 0;  // tasks whose control law wrote shadow integrator rows this cycle (lo: position / orientation / joint, hi: force / moment)
	auto commit_integrators = [&]() {  // workgroup-uniform; call after a barrier that follows the last shadow store
		for (int t = t_first; t < t_last; t++) {
			const TaskDev& tk = P.tasks[t];
			const int rows = tk.type == TASK_MOTION_FORCE ? 6 : tk.m;
			if ((commit_lo >> t) & 1u)
				for (int i = tid; i < rows; i += T) tk.integ[(size_t)i * P.ld + b] = tk.integ_new[(size_t)i * P.ld + b];
			if ((commit_hi >> t) & 1u)
				for (int i = tid; i < 6; i += T) tk.integ[(size_t)(6 + i) * P.ld + b] = tk.integ_new[(size_t)(6 + i) * P.ld + b];
		}
	};
	bool singular = false;
	bool truncated = false;  // some motion-force task was reduced to its non-singular subspace (singularity handling disabled)
	bool blended = false;    // some motion-force task ran the blended type-1 / type-2 singularity strategies
	MARK();
	// ---------------------------------------------------------------- task hierarchy
	for (int t = t_first; t < t_last; t++) {
		const TaskDev& tk = P.tasks[t];
		MARK();
		double* Jh = sm.W[4];
		int r = 0;           // rank of the task in its range basis
		bool empty = false;  // JointTask with no controllable dof left
		bool u_identity = true;
		bool passthrough = false;  // fully singular motion-force task with the strategies on: N = N_prec, zero torques
		bool mf_trunc = false;  // this motion-force task acts on the top-r left singular directions only; W[3] = their k x r coordinates
		bool blend = false;     // this motion-force task has r < k non-singular directions and the blended strategies are on
		const bool strategies = tk.type == TASK_MOTION_FORCE && tk.sing_handling && tk.sing_strategies && tk.sh;
		const int gs = tk.goal_comps;
		for (int e = tid; e < gs; e += T) sm.goal[e] = tk.goal[(size_t)e * P.ld + b];
		MARK();

		if (tk.type == TASK_MOTION_FORCE) {
			const int k = tk.k, bd = tk.body;
			// control point, control frame, world Jacobian [Jv; Jw] (JWorldFrame)
			const double* Rb = fk + 12 * bd;
			const double* ob = Rb + 9;
			double pw[3];
			mat3_mul_vec(Rb, tk.pos, pw);
			for (int e = 0; e < 3; e++) pw[e] += ob[e];
			if (tid < 3) sm.xp[tid] = pw[tid];
			if (tid < 9) {
				int rr = tid / 3, cc = tid - 3 * rr;
				sm.Rc[tid] = Rb[3 * rr] * tk.rot[cc] + Rb[3 * rr + 1] * tk.rot[3 + cc] + Rb[3 * rr + 2] * tk.rot[6 + cc];
			}
			for (int j = tid; j < n; j += T) {
				double col[6] = {0, 0, 0, 0, 0, 0};
				if (j <= bd) {
					const double* z = sm.zax + 3 * j;
					if (md.jtype[j] == 1) {
						const double* o = fk + 12 * j + 9;
						double rx = pw[0] - o[0], ry = pw[1] - o[1], rz = pw[2] - o[2];
						col[0] = z[1] * rz - z[2] * ry;
						col[1] = z[2] * rx - z[0] * rz;
						col[2] = z[0] * ry - z[1] * rx;
						col[3] = z[0]; col[4] = z[1]; col[5] = z[2];
					} else {
						col[0] = z[0]; col[1] = z[1]; col[2] = z[2];
					}
				}
				for (int e = 0; e < 6; e++) sm.Jw[e * LD + j] = col[e];
			}
			SYNC();
			// Jb = Bm^T Jw (k x n): rows of the task Jacobian in the basis of range(P)
			gemm<T>(sm.W[0], LD, tk.Bm, 1, 6, sm.Jw, LD, 1, k, 6, n, tid);
			// task-frame velocities use the UNprojected J = P Jw (MotionForceTask.cpp:293-298): vw = Jw dq
			if (tid < 6) {
				double s = 0.0;
				for (int j = 0; j < n; j++) s = fma(sm.Jw[tid * LD + j], sm.dq[j], s);
				sm.vw[tid] = s;
			}
			SYNC();
			gemm<T>(Jh, LD, sm.W[0], LD, 1, Np, LD, 1, k, n, n, tid);  // Jh = Jb N_prec
			SYNC();
			// singular values of the projected Jacobian: eig(Jh Jh^T) = sigma^2   (SingularityHandler.cpp:78-121)
			gemm<T>(sm.W[1], LD, Jh, LD, 1, Jh, 1, LD, k, n, k, tid);
			SYNC();
			// fast path: certify "sigma_i/sigma_0 >= s_max for all i and sigma_0 >= s_abs_tol" without an eigen-solve
			// (trace-power bound lambda_max <= ||Gs^4||_F^(1/4) + LDL^T pivots of Gs - s_max^2 u I; see saip_kernel_lane.hip)
			bool sing = false;
			{
				double tr = 0.0;
				for (int i = 0; i < k; i++) tr += sm.W[1][i * LD + i];
				const bool nonfinite = !(tr - tr == 0.0);  // NaN or infinite state / goal: trace(Jh Jh^T) is not a number
				const double itr = 1.0 / tr;
				double* Ga = sm.W[5];
				double* Gb = sm.W[6];
				for (int e = tid; e < k * k; e += T) {
					int i = e / k, j = e - i * k;
					Ga[i * LD + j] = sm.W[1][i * LD + j] * itr;
				}
				SYNC();
				for (int rep = 0; rep < 2; rep++) {
					gemm<T>(Gb, LD, Ga, LD, 1, Ga, LD, 1, k, k, k, tid);
					SYNC();
					double* t_ = Ga;
					Ga = Gb;
					Gb = t_;
				}
				double t8 = 0.0;
				for (int i = 0; i < k; i++)
					for (int j = 0; j < k; j++) t8 = fma(Ga[i * LD + j], Ga[i * LD + j], t8);
				const double u = sqrt(sqrt(sqrt(t8)));
				const bool ok_s0 = (u * tk.cert_kroot * tr >= tk.s_abs_tol * tk.s_abs_tol);
				if (tid == 0) {  // LDL^T of Gs - shift I (k <= 6: one thread, in registers; rows >= k padded with the identity)
					const double shift = tk.s_max * tk.s_max * u;
					double Gs[6][6];
#pragma unroll
					for (int i = 0; i < 6; i++)
#pragma unroll
						for (int j = 0; j <= i; j++)
							Gs[i][j] = (i < k) ? sm.W[1][i * LD + j] * itr - (i == j ? shift : 0.0) : (i == j ? 1.0 : 0.0);
					int pos = 1;
#pragma unroll
					for (int kk = 0; kk < 6; kk++) {
						const double d = Gs[kk][kk];
						if (!(d > 1e-13)) pos = 0;
						const double id = wg_rcp(d);  // (a non-positive pivot has already failed the certificate: whatever follows is not used)
#pragma unroll
						for (int i = kk + 1; i < 6; i++) {
							const double lik = Gs[i][kk] * id;
#pragma unroll
							for (int j = kk + 1; j <= i; j++) Gs[i][j] = fma(-lik, Gs[j][kk], Gs[i][j]);
						}
					}
					sm.flag = (pos && ok_s0 && tr > 0.0) ? 1 : 0;
				}
				SYNC();
				const bool certified = sm.flag != 0;
				SYNC();
				r = k;
				if (!certified) {  // exact path: singular values from the Gram eigenvalues (SingularityHandler.cpp:78-121)
					jacobi_eig<T>(sm.W[1], (tk.sing_handling && !strategies) ? nullptr : sm.W[2], k, LD, sm.cs, &sm.flag, tid);
					double lmax = 0.0;
					for (int i = 0; i < k; i++) lmax = fmax(lmax, sm.W[1][i * LD + i]);
					double s0 = sqrt(lmax);
					if (tk.sing_handling && !strategies) {
						// handling strategy enforced (reference default) without the blended strategies switched on -> flagged
						sing = !(s0 >= tk.s_abs_tol);
						for (int i = 0; i < k; i++) {
							double si = sqrt(fmax(sm.W[1][i * LD + i], 0.0));
							if (si / s0 < tk.s_max) sing = true;
						}
					} else {
						// disableSingularityHandling(): _N = _N_ns and tau = tau_ns (SingularityHandler.cpp:146-148, 319-330): the task acts on
						// the span of the leading left singular vectors U[:, :i], i = first index with sigma_i / sigma_0 < s_max (:100-121)
						if (tid == 0) {
							int ord[6];
							for (int i = 0; i < k; i++) ord[i] = i;
							for (int i = 1; i < k; i++)  // insertion sort, descending eigenvalue
								for (int j = i; j > 0 && sm.W[1][ord[j] * LD + ord[j]] > sm.W[1][ord[j - 1] * LD + ord[j - 1]]; j--) {
									int t_ = ord[j];
									ord[j] = ord[j - 1];
									ord[j - 1] = t_;
								}
							int ns = k;
							if (!(s0 >= tk.s_abs_tol)) ns = 0;  // fully singular (:83-98): the task is passed through (tau = 0, :316-317)
							else
								for (int i = 1; i < k; i++)
									if (sqrt(fmax(sm.W[1][ord[i] * LD + ord[i]], 0.0)) / s0 < tk.s_max) {
										ns = i;
										break;
									}
							sm.rank = ns;
							for (int i = 0; i < k; i++) sm.colidx[i] = ord[i];
						}
						SYNC();
						r = sm.rank;
						if (strategies) {
							if (r < k) {
								// V_s = Jp^T U_s / sigma_s needs sigma_s well above the rounding noise of the Gram eigenvalues; a fully singular
								// task (sigma_0 < 1e-3) or sigma_min / sigma_0 < 1e-6 is reported as singular instead
								double lmin = lmax;
								for (int i = 0; i < k; i++) lmin = fmin(lmin, sm.W[1][i * LD + i]);
								if (nonfinite) {
									sing = true;  // non-finite state: refused, never passed through
								} else if (r == 0) {
									// fully singular task (sigma_0 < s_abs_tol): passed through like in the reference -- zero torques (:316-317) and
									// _N = N_prec (:149-151); its classification (garbage directions at this scale) is skipped
									empty = true;
									passthrough = true;
									truncated = true;
								} else {
									blend = true;  // (refused further down if a singular value turns out below 1e-10 sigma_0)
								}
								(void)lmin;
							}
						} else if (nonfinite) {
							sing = true;  // non-finite state: refused, never reduced or passed through
						} else if (r < k) {
							truncated = true;
							if (r == 0) {
								empty = true;
							} else {
								mf_trunc = true;
								for (int e = tid; e < k * r; e += T) {
									int i = e / r, c = e - i * r;
									sm.W[3][i * LD + c] = sm.W[2][i * LD + sm.colidx[c]];  // U_ns in range(P) coordinates
								}
								SYNC();
								gemm<T>(sm.W[0], LD, sm.W[3], 1, LD, Jh, LD, 1, r, k, n, tid);  // J_ns = U_ns^T Jh
								SYNC();
								for (int e = tid; e < r * n; e += T) {
									int i = divn(e), j = e - i * n;
									Jh[i * LD + j] = sm.W[0][i * LD + j];
								}
							}
						}
					}
				}
			}
			if (sing) singular = true;
			// (no singular directions: SingularityHandler.cpp:238-244 clears the handler's counters; here they are void by construction
			// unless the previous cycle was blended too -- ShState::last_cycle -- so nothing is written for such instances)
			SYNC();
		} else {
			const int m = tk.m;
			// Jp = S N_prec  (JointTask.cpp:231)
			const double* Jp = Np;
			if (!tk.s_identity) {
				gemm<T>(sm.W[0], LD, tk.S, n, 1, Np, LD, 1, m, n, n, tid);
				Jp = sm.W[0];
				SYNC();
			}
			// matrixRangeBasis(Jp) (JointTask.cpp:233).  Fast path: column-pivoted Gram-Schmidt in LDS with a clean-gap
			// acceptance rule (next pivot below 1e-10 or above 1e-2 of the first); the deflation coefficients are the rows of U^T Jp.
			bool need_exact = false;
			{
				// Thread (j, g) = (tid % NMAX, tid / NMAX) owns column j, rows g, g+G, ...: column norms and deflation dot products are
				// G-way partial sums through LDS, the arg-max is a butterfly inside each NMAX-lane group (every thread gets the same
				// pivot, so no broadcast barrier) -> two barriers per accepted column.
				constexpr int G = T / NMAX;
				double* Wc = sm.W[5];
				double* pn = sm.W[1];            // [G][NMAX] partial squared column norms
				double* pd = sm.W[1] + T;        // [G][NMAX] partial deflation dot products
				const int rb = P.rank_bound[t] < m ? P.rank_bound[t] : m;
				const int cj = tid % NMAX, cg = tid / NMAX;
				{
					double acc = 0.0;
					if (cj < n)
						for (int i = cg; i < m; i += G) {
							const double v = Jp[i * LD + cj];
							Wc[i * LD + cj] = v;
							acc = fma(v, v, acc);
						}
					pn[tid] = acc;
				}
				SYNC();
				double c0 = 1.0;
				bool going = true;
				int rank = 0;
				for (int s = 0; s <= rb; s++) {
					double best = -1.0, frob2 = 0.0;
					int jb = cj;
					if (cj < n) {
						double a = 0.0;
						for (int g = 0; g < G; g++) a += pn[g * NMAX + cj];
						best = a;
						frob2 = a;
					}
					for (int off = 1; off < NMAX; off <<= 1) {
						const double ob = __shfl_xor(best, off);
						const int oj = __shfl_xor(jb, off);
						frob2 += __shfl_xor(frob2, off);
						if (ob > best || (ob == best && oj < jb)) {
							best = ob;
							jb = oj;
						}
					}
					if (s == 0) {
						c0 = best;
						if (frob2 < 1e-6) going = false;  // ||Jp||_F < 1e-3: empty range
						else if (best < 1e-5 || rb == 0) { going = false; need_exact = true; }  // (rb == 0: the host's rank bound assumed full-rank
						                                                                        // tasks above; one was passed through -- the exact path decides)
					} else if (going) {
						const double ratio = best * wg_rcp(c0);
						if (ratio < 1e-20) going = false;
						else if (ratio < 1e-4 || s == rb) { going = false; need_exact = true; }
					}
					if (!going || s == rb) break;
					const double inv = wg_rsqrt(best);
					{
						double acc = 0.0;
						if (cj < n)
							for (int i = cg; i < m; i += G) acc = fma(Wc[i * LD + jb], Wc[i * LD + cj], acc);
						pd[tid] = acc * inv;
					}
					if (tid < m) {
						const double u = Wc[tid * LD + jb] * inv;
						sm.W[3][tid * LD + s] = u;  // U[:, s]
						sm.v1[tid] = u;
					}
					SYNC();
					{
						double acc = 0.0;
						if (cj < n) {
							double d = 0.0;
							for (int g = 0; g < G; g++) d += pd[g * NMAX + cj];
							if (cg == 0) Jh[s * LD + cj] = d;  // row s of U^T Jp
							for (int i = cg; i < m; i += G) {
								const double v = fma(-sm.v1[i], d, Wc[i * LD + cj]);
								Wc[i * LD + cj] = v;
								acc = fma(v, v, acc);
							}
						}
						pn[tid] = acc;
					}
					SYNC();
					rank = s + 1;
				}
				r = rank;
				empty = (r == 0) && !need_exact;
				u_identity = false;
			}
			if (need_exact) {
			// exact path: eigen-decomposition of the Gram matrix Jp Jp^T (m x m), reference tolerances
			gemm<T>(sm.W[1], LD, Jp, LD, 1, Jp, 1, LD, m, n, m, tid);
			SYNC();
			double tr = 0.0;
			for (int i = 0; i < m; i++) tr += sm.W[1][i * LD + i];  // ||Jp||_F^2
			SYNC();
			jacobi_eig<T>(sm.W[1], sm.W[2], m, LD, sm.cs, &sm.flag, tid);
			if (tid == 0) {
				double lmax = 0.0;
				for (int i = 0; i < m; i++) lmax = fmax(lmax, sm.W[1][i * LD + i]);
				int rr = 0;
				if (sqrt(fmax(tr, 0.0)) >= 1e-3 && sqrt(lmax) >= 1e-3) {
					for (int i = 0; i < m; i++)
						if (sqrt(fmax(sm.W[1][i * LD + i], 0.0) / lmax) >= 1e-3) sm.colidx[rr++] = i;
				}
				sm.rank = rr;
			}
			SYNC();
			r = sm.rank;
			empty = (r == 0);
			u_identity = (r == m);
			if (!empty) {
				if (u_identity) {
					for (int e = tid; e < m * n; e += T) {
						int i = divn(e), j = e - i * n;
						Jh[i * LD + j] = Jp[i * LD + j];
					}
				} else {
					// U = selected eigenvectors (m x r) -> W[3];  Jh = U^T Jp
					for (int e = tid; e < m * r; e += T) {
						int i = e / r, c = e - i * r;
						sm.W[3][i * LD + c] = sm.W[2][i * LD + sm.colidx[c]];
					}
					SYNC();
					gemm<T>(Jh, LD, sm.W[3], 1, LD, Jp, LD, 1, r, m, n, tid);
				}
			}
			}
			SYNC();
		}

		double* Lam = sm.W[2];
		double* Lmod = sm.W[7];
		double* Nt = sm.W[6];
		// operationalSpaceMatrices(J) (rows x n): Lambda = (J M^-1 J^T)^-1 -> Lam, N = I - M^-1 J^T Lambda J -> Nt (when asked for),
		// Lambda_modified by the task's decoupling type -> Lmod (SingularityHandler.cpp:175-193).  Scratch: W[0], W[1], W[5], W[6].
		auto osm = [&](const double* J, int rows, bool want_N) {
			gemm<T>(sm.W[0], LD, J, LD, 1, sm.Minv, LD, 1, rows, n, n, tid);  // T1 = J Minv
			SYNC();
			gemm<T>(sm.W[1], LD, sm.W[0], LD, 1, J, 1, LD, rows, n, rows, tid);  // A = T1 J^T
			SYNC();
			spd_inverse<T>(sm.W[1], Lam, sm.W[5], sm.W[6], sm.dv, rows, LD, tid);
			if (want_N) {
				gemm<T>(sm.W[5], LD, Lam, LD, 1, sm.W[0], LD, 1, rows, rows, n, tid);  // T2 = Lambda T1 = Jbar^T
				SYNC();
				gemm<T>(Nt, LD, sm.W[5], 1, LD, J, LD, 1, n, rows, n, tid, GEMM_I_MINUS);  // N = I - T2^T J
				SYNC();
			}
			Lmod = sm.W[7];
			if (tk.decoupling == DEC_FULL) {
				Lmod = Lam;
			} else if (tk.decoupling == DEC_IMPEDANCE) {
				for (int e = tid; e < rows * rows; e += T) {
					int i = e / rows, j = e - i * rows;
					Lmod[i * LD + j] = (i == j) ? 1.0 : 0.0;
				}
				SYNC();
			} else {
				if (tk.bie_threshold != cur_bie_thr) {  // another threshold than the cached one: rebuild M_BIE^-1
					if (!bie_inverse_lowrank<T>(Mkeep, sm.Minv, sm.MBinv, sm.bieK, sm.bieIdx, sm.W[0], tk.bie_threshold, n, LD, tid)) {
						for (int e = tid; e < n * n; e += T) {
							int i = divn(e), j = e - i * n;
							double v = Mkeep[i * LD + j];
							if (i == j && v < tk.bie_threshold) v = tk.bie_threshold;
							sm.W[1][i * LD + j] = v;
						}
						SYNC();
						spd_inverse<T>(sm.W[1], sm.MBinv, sm.W[0], sm.W[5], sm.dv, n, LD, tid);
					}
					cur_bie_thr = tk.bie_threshold;
				}
				gemm<T>(sm.W[0], LD, J, LD, 1, sm.MBinv, LD, 1, rows, n, n, tid);
				SYNC();
				gemm<T>(sm.W[1], LD, sm.W[0], LD, 1, J, 1, LD, rows, n, rows, tid);
				SYNC();
				spd_inverse<T>(sm.W[1], Lmod, sm.W[5], sm.W[0], sm.dv, rows, LD, tid);
			}
		};
		const bool track = (tk.has_ki || P.integ_always);
		// MotionForceTask::computeTorques up to the task forces: F_um -> sm.Fum, F_f -> sm.Ff, their range(P) coordinates -> v0, v1
		auto mf_law = [&]() {
			if (tid == 0) {
				const auto& G = sm.goal;  // x3, R9, v3, w3, a3, alpha3, f3, m3, sensed f3, sensed m3
				double ip[3] = {0, 0, 0}, io[3] = {0, 0, 0};
				if (track)
					for (int i = 0; i < 3; i++) {
						ip[i] = tk.integ[(size_t)i * P.ld + b];
						io[i] = tk.integ[(size_t)(3 + i) * P.ld + b];
					}
				// general law (velocity saturation, force/moment spaces) reduces to MotionForceTask.cpp:411-468 for the defaults
				double ifm[6] = {0, 0, 0, 0, 0, 0};
				const bool closed = tk.cl_force || tk.cl_moment;
				if (closed)
					for (int i = 0; i < 6; i++) ifm[i] = tk.integ[(size_t)(6 + i) * P.ld + b];
				bool ovf = false;
				law_motion_force_general<true>(&tk, sm.vw, sm.xp, sm.Rc, G, ip, io, ifm, sm.Fum, sm.Ff, tk.popc ? tk.popc + b : nullptr, P.ld, &ovf);
				if (ovf) sm.flag2 = 1;
				// shadow copies: committed at the end of the cycle unless the instance ends flagged (a later task may still flag it)
				if (closed)
					for (int i = 0; i < 6; i++) tk.integ_new[(size_t)(6 + i) * P.ld + b] = ifm[i];
				if (track)
					for (int i = 0; i < 3; i++) {
						tk.integ_new[(size_t)i * P.ld + b] = ip[i];
						tk.integ_new[(size_t)(3 + i) * P.ld + b] = io[i];
					}
			}
			if (track) commit_lo |= 1u << t;
			if (tk.cl_force || tk.cl_moment) commit_hi |= 1u << t;
			SYNC();
			if (tid < tk.k) {  // fhat = Bm^T F_um, ffhat = Bm^T F_f
				double s = 0.0, s2 = 0.0;
				for (int i = 0; i < 6; i++) {
					s = fma(tk.Bm[i * 6 + tid], sm.Fum[i], s);
					s2 = fma(tk.Bm[i * 6 + tid], sm.Ff[i], s2);
				}
				sm.v0[tid] = s;
				sm.v1[tid] = s2;
			}
			SYNC();
		};
		if (blend && !singular) {
			// ------------------------------------------------------------ blended singularity strategies, SingularityHandler.cpp:100-121,
			// 146-228 (model), 230-295 (classification), 310-367 (torques); sign conventions as DESIGN.md states.
			// Everything in range(P) coordinates: U (k x k, sorted by singular value) = eigenvectors of Jh Jh^T, V_s^T = diag(1/sigma_s) U_s^T Jh.
			blended = true;
			const int k = tk.k, ns = r, nsg = k - r;
			ShState& st = tk.sh[b];
			if (do_state) {  // counters and window of an instance that had no singular direction in the previous cycle are void (:238-244)
				if (tid == 0 && st.last_cycle != P.task_cycle[t] - 1) st.ntypes = st.c1 = st.c2 = st.head = st.size = 0;
				SYNC();
			}
			double* tns = sm.Jw;            // Jw is dead once Jb and vw exist: rows 0..4 = tau_ns, tau_s, tau_js, wA, wB
			double* ts = sm.Jw + LD;
			double* wA = sm.Jw + 3 * LD;
			double* wB = sm.Jw + 4 * LD;
			for (int e = tid; e < k * k; e += T) {
				int i = e / k, c = e - i * k;
				sm.W[3][i * LD + c] = sm.W[2][i * LD + sm.colidx[c]];
			}
			if (tid < k) sm.dv[tid] = sqrt(fmax(sm.W[1][sm.colidx[tid] * LD + sm.colidx[tid]], 0.0));
			SYNC();
			gemm<T>(sm.W[0], LD, sm.W[3], 1, LD, Jh, LD, 1, k, k, n, tid);  // rows 0..ns-1: J_ns = U_ns^T Jh, rows ns..k-1: J_s = U_s^T Jh
			SYNC();
			for (int e = tid; e < k * n; e += T) {
				int i = divn(e), j = e - i * n;
				Jh[i * LD + j] = sm.W[0][i * LD + j];
			}
			SYNC();
			// singular values of the singular directions from the rows themselves, sigma_c = |u_c^T Jh| (then V_s = row / sigma_c): accurate
			// down to sigma_c / sigma_0 ~ 1e-10, where the Gram eigenvalue (sigma^2 against lambda_max) has long drowned in rounding noise
			if (tid >= ns && tid < k) {
				double s2 = 0.0;
				for (int j = 0; j < n; j++) s2 = fma(Jh[tid * LD + j], Jh[tid * LD + j], s2);
				sm.dv[tid] = sqrt(s2);
			}
			SYNC();
			bool too_small = false;  // workgroup-uniform
			for (int c = ns; c < k; c++)
				if (!(sm.dv[c] >= 1e-10 * sm.dv[0])) too_small = true;
			if (too_small) {  // the direction of V_s is not resolved in double precision: refused (status 1), never guessed
				singular = true;
				blended = false;
			} else {
			const double alpha = fmin(fmax((sm.dv[ns] / sm.dv[0] - tk.s_min) / (tk.s_max - tk.s_min), 0.0), 1.0);  // :114-115
			double fTd = 0.0;
			if (do_law) {
				mf_law();
				double s = 0.0, s2 = 0.0;
				if (tid < k)
					for (int i = 0; i < k; i++) {
						s = fma(sm.W[3][i * LD + tid], sm.v0[i], s);
						s2 = fma(sm.W[3][i * LD + tid], sm.v1[i], s2);
					}
				SYNC();
				if (tid < k) {
					sm.v2[tid] = s;   // U^T F_um
					sm.v4[tid] = s2;  // U^T F_f
				}
				SYNC();
				double fn2 = 0.0;
				for (int i = 0; i < 6; i++) fn2 = fma(sm.Fum[i] + sm.Ff[i], sm.Fum[i] + sm.Ff[i], fn2);
				fTd = (sm.v2[ns] + sm.v4[ns]) / sqrt(fn2);  // :346-347
			}
			// non-singular part (:104-112): Lambda_ns, N_ns, tau_ns (:321-322)
			osm(Jh, ns, true);
			if (do_law) {
				if (tid < ns) {
					double s = sm.v4[tid];
					for (int i = 0; i < ns; i++) s = fma(Lmod[tid * LD + i], sm.v2[i], s);
					sm.v3[tid] = s;
				}
				SYNC();
				for (int j = tid; j < n; j += T) {
					double s = 0.0;
					for (int c = 0; c < ns; c++) s = fma(Jh[c * LD + j], sm.v3[c], s);
					tns[j] = s;
				}
			}
			const bool want_task_N = P.out_N || (P.diag && tk.diag_N);
			if (want_task_N)
				for (int e = tid; e < n * n; e += T) {
					int i = divn(e), j = e - i * n;
					sm.Nx[i * LD + j] = Nt[i * LD + j];  // N_ns, for getTaskNullspace = N_js N_ns
				}
			gemm<T>(Np2, LD, Nt, LD, 1, Np, LD, 1, n, n, n, tid);  // X = N_ns N_prec
			SYNC();
			// singular part (:116-121): Lambda_s, tau_s clamped to the torque limits (:358-366)
			if (do_law) {
				osm(Jh + ns * LD, nsg, false);
				if (tid < nsg) {
					double s = sm.v4[ns + tid];
					for (int i = 0; i < nsg; i++) s = fma(Lmod[tid * LD + i], sm.v2[ns + i], s);
					sm.v3[tid] = s;
				}
				SYNC();
				for (int j = tid; j < n; j += T) {
					double s = 0.0;
					for (int c = 0; c < nsg; c++) s = fma(Jh[(ns + c) * LD + j], sm.v3[c], s);
					const double lim = md.effort[j];
					ts[j] = (s != s) ? 0.0 : fmin(fmax(s, -lim), lim);
				}
			}
			// classifySingularity (:230-295): task-space motion along u_c of the control frame at q +- 5 v_c
			double* qp = sm.W[5];  // [2 nsg][n] perturbed postures (W[5] and W[6] are contiguous)
			for (int e = tid; e < 2 * nsg * n; e += T) {
				int pidx = divn(e), j = e - pidx * n, c = pidx >> 1;
				qp[e] = sm.q[j] + ((pidx & 1) ? -5.0 : 5.0) * (Jh[(ns + c) * LD + j] / sm.dv[ns + c]);
			}
			if (tid < n) {  // :232-235 with the counters of the previous cycle
				double qpr = st.q_prior[tid];
				if (do_state && (st.ntypes == 0 || st.c2 > st.c1)) {
					qpr = sm.q[tid];
					st.q_prior[tid] = qpr;
				}
				sm.v0[tid] = qpr;
			}
			SYNC();
			if (do_state && tid < 2 * nsg) {
				double x1[3], R1[9], w6[6] = {0, 0, 0, 0, 0, 0};
				fk_control_frame(md, tk, qp + tid * n, 1, 0, x1, R1);
				for (int e = 0; e < 3; e++) w6[e] = x1[e] - sm.xp[e];
				for (int col = 0; col < 3; col++) {  // orientation error of R1 with respect to Rc
					const double a0 = sm.Rc[col], a1 = sm.Rc[3 + col], a2 = sm.Rc[6 + col];
					const double b0 = R1[col], b1 = R1[3 + col], b2 = R1[6 + col];
					w6[3] -= 0.5 * (a1 * b2 - a2 * b1);
					w6[4] -= 0.5 * (a2 * b0 - a0 * b2);
					w6[5] -= 0.5 * (a0 * b1 - a1 * b0);
				}
				const int c = tid >> 1;
				double d = 0.0;
				for (int i = 0; i < k; i++) {
					double bi = 0.0;
					for (int e = 0; e < 6; e++) bi = fma(tk.Bm[e * 6 + i], w6[e], bi);
					d = fma(bi, sm.W[3][i * LD + ns + c], d);
				}
				sm.cs[tid] = fabs(d);
			}
			SYNC();
			if (tid == 0) {
				int c1 = st.c1, c2 = st.c2;
				if (do_state) {
					bool any1 = false;
					for (int c = 0; c < nsg; c++)
						if (fmax(sm.cs[2 * c], sm.cs[2 * c + 1]) > 0.5) any1 = true;  // _type_1_tol
					st.ntypes = nsg;
					int head = st.head, size = st.size;
					if (size == SH_BUFFER) {  // :285-292
						if (st.hist[head] == 1) c1--;
						else c2--;
						head = (head + 1) % SH_BUFFER;
						size--;
					}
					st.hist[(head + size) % SH_BUFFER] = any1 ? 1 : 2;
					size++;
					if (any1) c1++;
					else c2++;
					st.head = head;
					st.size = size;
					st.c1 = c1;
					st.c2 = c2;
					st.last_cycle = P.task_cycle[t];
				}
				unsigned dn = st.dir_neg;
				const bool t1 = (c1 > c2) || tk.sh_force_type1;  // :329
				if (!t1 && do_law) {  // type-2 strategy: torque direction flips near the joint limits (:339-345; inside computeTorques)
					for (int j = 0; j < n; j++)
						if (Jh[ns * LD + j] != 0.0) {
							if (fabs(sm.q[j] - md.q_upper[j]) < 5.0 * M_PI / 180.0) dn |= 1u << j;
							else if (fabs(sm.q[j] - md.q_lower[j]) < 5.0 * M_PI / 180.0) dn &= ~(1u << j);
						}
					st.dir_neg = dn;
				}
				sm.flag = t1 ? 1 : 0;
				sm.rank = (int)dn;
			}
			SYNC();
			const bool type1 = sm.flag != 0;
			const unsigned dir_neg = (unsigned)sm.rank;
			if (do_law && tid < n) {
				if (type1) {  // hold the entering posture (:333-336)
					sm.v1[tid] = -tk.sh_kp1 * (sm.q[tid] - sm.v0[tid]) - tk.sh_kv1 * sm.dq[tid];
				} else {      // :346-354
					sm.v1[tid] = (((dir_neg >> tid) & 1u) ? -1.0 : 1.0) * (fabs(fTd) * 1e-2 * md.effort[tid]);
					sm.v2[tid] = -tk.sh_kv2 * sm.dq[tid];
				}
			}
			SYNC();
			if (do_law && tid < nsg) {  // V_s^T unit
				double a = 0.0, b2 = 0.0;
				for (int j = 0; j < n; j++) {
					a = fma(Jh[(ns + tid) * LD + j], sm.v1[j], a);
					if (!type1) b2 = fma(Jh[(ns + tid) * LD + j], sm.v2[j], b2);
				}
				wA[tid] = a / sm.dv[ns + tid];
				wB[tid] = b2 / sm.dv[ns + tid];
			}
			// joint strategy: posture Jacobian V_s^T N_ns N_prec (:153), Lambda_js, N_js
			gemm<T>(sm.W[0], LD, Jh + ns * LD, LD, 1, Np2, LD, 1, nsg, n, n, tid);
			SYNC();
			for (int e = tid; e < nsg * n; e += T) {
				int c = divn(e), j = e - c * n;
				Jh[c * LD + j] = sm.W[0][c * LD + j] / sm.dv[ns + c];
			}
			SYNC();
			osm(Jh, nsg, true);
			if (do_law) {
				if (tid < nsg) {
					double s = type1 ? 0.0 : wA[tid];
					for (int i = 0; i < nsg; i++) s = fma(Lmod[tid * LD + i], type1 ? wA[i] : wB[i], s);
					sm.v3[tid] = s;
				}
				SYNC();
				for (int j = tid; j < n; j += T) {
					double s = 0.0;
					for (int c = 0; c < nsg; c++) s = fma(Jh[c * LD + j], sm.v3[c], s);
					if (tk.decoupling == DEC_IMPEDANCE) sm.tau[j] += tns[j];  // :311-313
					else sm.tau[j] += tns[j] + alpha * ts[j] + (1.0 - alpha) * s;  // :367
				}
			}
			gemm<T>(Np, LD, Nt, LD, 1, Np2, LD, 1, n, n, n, tid);  // N_prec <- N_js N_ns N_prec (:157 + getTaskAndPreviousNullspace)
			SYNC();
			if (want_task_N) {  // _N = N_js N_ns (:157)
				gemm<T>(Np2, LD, Nt, LD, 1, sm.Nx, LD, 1, n, n, n, tid);
				SYNC();
				if (P.diag && tk.diag_N)
					for (int e = tid; e < n * n; e += T) {
						int i = divn(e), j = e - i * n;
						tk.diag_N[(size_t)e * P.ld + b] = Np2[i * LD + j];
					}
			}
			if (P.single_task >= 0 && (P.out_N || P.out_Ntot)) dump_nullspaces(Np2, Np);
			}  // !too_small
		} else if (!empty && !singular) {
			// the nullspace of the last task of a hierarchy feeds nothing (unless somebody asked to see it)
			const bool need_N = (t + 1 < t_last) || (P.diag && tk.diag_N) || P.out_N || P.out_Ntot;
			osm(Jh, r, need_N);
			if (P.diag && tk.diag_N) {
				for (int e = tid; e < n * n; e += T) {
					int i = divn(e), j = e - i * n;
					tk.diag_N[(size_t)e * P.ld + b] = Nt[i * LD + j];
				}
			}

			// ------------------------------------------------------------ control law -> g (r) in v3, then tau_t = Jh^T g
			if (!do_law) {
				// updateTaskModel only: the nullspaces are all that is asked for
			} else if (tk.type == TASK_MOTION_FORCE) {
				mf_law();
				if (mf_trunc) {  // U_ns^T of both (task_range_ns^T F, SingularityHandler.cpp:321-322)
					double s = 0.0, s2 = 0.0;
					if (tid < r)
						for (int i = 0; i < tk.k; i++) {
							s = fma(sm.W[3][i * LD + tid], sm.v0[i], s);
							s2 = fma(sm.W[3][i * LD + tid], sm.v1[i], s2);
						}
					SYNC();
					if (tid < r) {
						sm.v0[tid] = s;
						sm.v1[tid] = s2;
					}
					SYNC();
				}
				if (tid < r) {  // g = Lambda_mod fhat + ffhat   (SingularityHandler.cpp:307-309)
					double s = sm.v1[tid];
					for (int i = 0; i < r; i++) s = fma(Lmod[tid * LD + i], sm.v0[i], s);
					sm.v3[tid] = s;
				}
				SYNC();
				// computeTorques(tau_prec): the reference's _Lambda is never written => no compensation
			} else {
				const int m = tk.m;
				// e, f (m-vectors)
				if (track) commit_lo |= 1u << t;
				for (int i = tid; i < m; i += T) {
					double cur = 0.0, vel = 0.0;
					if (tk.s_identity) {
						cur = sm.q[i];
						vel = sm.dq[i];
					} else {
						for (int j = 0; j < n; j++) {
							cur = fma(tk.S[i * n + j], sm.q[j], cur);
							vel = fma(tk.S[i * n + j], sm.dq[j], vel);
						}
					}
					double e = cur - sm.goal[i];
					double ie = track ? tk.integ[(size_t)i * P.ld + b] : 0.0;
					ie += e * tk.dt;  // JointTask.cpp:323-324
					if (track) tk.integ_new[(size_t)i * P.ld + b] = ie;
					if (tk.vel_sat) {  // JointTask.cpp:327-341
						double vdes = -tk.kp[i] * tk.kvinv[i] * e - tk.ki[i] * tk.kvinv[i] * ie;
						vdes = fmin(fmax(vdes, -tk.sat[i]), tk.sat[i]);
						sm.v0[i] = -tk.kv[i] * (vel - vdes);
					} else {
						sm.v0[i] = -tk.kp[i] * e - tk.kv[i] * (vel - sm.goal[m + i]) - tk.ki[i] * ie;  // :342-345
					}
				}
				// w = Minv tau_prec
				for (int i = tid; i < n; i += T) {
					double s = 0.0;
					for (int j = 0; j < n; j++) s = fma(sm.Minv[i * LD + j], sm.tau[j], s);
					sm.v1[i] = s;
				}
				SYNC();
				// sw = S w - (nothing): v2 (m)
				for (int i = tid; i < m; i += T) {
					double s = 0.0;
					if (tk.s_identity) s = sm.v1[i];
					else
						for (int j = 0; j < n; j++) s = fma(tk.S[i * n + j], sm.v1[j], s);
					sm.v2[i] = s;
				}
				SYNC();
				// range coordinates: a1 = U^T ddq_d - U^T sw (acts through Lambda), b1 = U^T f (acts through Lambda_mod)
				for (int c = tid; c < r; c += T) {
					double a1 = 0.0, b1 = 0.0;
					if (u_identity) {
						a1 = sm.goal[2 * m + c] - sm.v2[c];
						b1 = sm.v0[c];
					} else {
						for (int i = 0; i < m; i++) {
							double u = sm.W[3][i * LD + c];
							a1 = fma(u, sm.goal[2 * m + i] - sm.v2[i], a1);
							b1 = fma(u, sm.v0[i], b1);
						}
					}
					sm.v4[c] = a1;
					sm.dv[c] = b1;
				}
				SYNC();
				for (int c = tid; c < r; c += T) {  // g = Lambda a1 + Lambda_mod b1   (JointTask.cpp:348-351, 287-291)
					double s = 0.0;
					for (int i = 0; i < r; i++) s = fma(Lam[c * LD + i], sm.v4[i], fma(Lmod[c * LD + i], sm.dv[i], s));
					sm.v3[c] = s;
				}
				SYNC();
			}
			if (do_law)
				for (int j = tid; j < n; j += T) {  // tau += Jh^T g
					double s = 0.0;
					for (int c = 0; c < r; c++) s = fma(Jh[c * LD + j], sm.v3[c], s);
					sm.tau[j] += s;
				}
			// N_prec <- N N_prec   (getTaskAndPreviousNullspace)
			if (need_N) {
				gemm<T>(Np2, LD, Nt, LD, 1, Np, LD, 1, n, n, n, tid);
				SYNC();
				double* t_ = Np;
				Np = Np2;
				Np2 = t_;
			}
			if (P.single_task >= 0 && (P.out_N || P.out_Ntot)) dump_nullspaces(Nt, Np);
		} else if (empty && passthrough) {
			// SingularityHandler.cpp:149-151: _N = N_prec, so the next task is updated with N_prec N_prec (= N_prec up to rounding for the
			// projectors of a hierarchy)
			gemm<T>(Np2, LD, Np, LD, 1, Np, LD, 1, n, n, n, tid);
			SYNC();
			if (P.diag && tk.diag_N)
				for (int e = tid; e < n * n; e += T) {
					int i = divn(e), j = e - i * n;
					tk.diag_N[(size_t)e * P.ld + b] = Np[i * LD + j];
				}
			if (P.single_task >= 0 && (P.out_N || P.out_Ntot)) dump_nullspaces(Np, Np2);
			double* t_ = Np;
			Np = Np2;
			Np2 = t_;
		} else if (empty) {
			// no controllable direction left: N = I (JointTask.cpp:234-239), a fully singular reduced motion-force task is passed through
			if (P.diag && tk.diag_N)
				for (int e = tid; e < n * n; e += T) {
					int i = divn(e), j = e - i * n;
					tk.diag_N[(size_t)e * P.ld + b] = (i == j) ? 1.0 : 0.0;
				}
			if (P.single_task >= 0 && (P.out_N || P.out_Ntot)) dump_nullspaces(nullptr, Np);
		}
		SYNC();
		if (singular) break;
	}

	if (P.single_task >= 0) {
		// TemplateTask::computeTorques[(tau_prec)]: this task's torques alone (sm.tau started from tau_prec: it feeds the joint task's
		// disturbance compensation, JointTask.cpp:285-292); none of RobotController's post-processing.  A flagged instance follows the flagged-torque policy (held or NaN).
		const double nanv = __builtin_nan("");
		if (tid < n && do_law) {
			const double tp = P.ext_tau_prec ? P.ext_tau_prec[(size_t)tid * P.ld + b] : 0.0;
			if (!singular) P.tau[(size_t)tid * P.ld + b] = sm.tau[tid] - tp;
			else if (P.flag_nan) P.tau[(size_t)tid * P.ld + b] = nanv;  // otherwise the torques stay what they were: the last valid ones are held
		}
		if (singular && !do_law)
			for (int e = tid; e < n * n; e += T) {
				if (P.out_N) P.out_N[(size_t)e * P.ld + b] = nanv;
				if (P.out_Ntot) P.out_Ntot[(size_t)e * P.ld + b] = nanv;
			}
		if (tid == 0) P.status[b] = singular ? 1 : ((truncated ? 2 : 0) | (sm.flag2 ? 4 : 0) | (blended ? 8 : 0));
		if (!singular) commit_integrators();
		return;
	}
	// ---------------------------------------------------------------- post-processing, RobotController.cpp:86-116
	if (tid < n && P.torque_sat) {
		const double lim = md.effort[tid];
		const double tv = sm.tau[tid];
		sm.tau[tid] = tv > lim ? lim : (tv < -lim ? -lim : tv);
	}
	if (P.jla && !singular) {
		// tau = JLA.computeTorques(tau) + N_constraints^T tau,  N_constraints^T = I - S^T (S M^-1 S^T)^-1 S M^-1 (padded to n x n)
		SYNC();
		if (tid < n) {
			bool act;
			sm.v0[tid] = jla_joint(sm.q[tid], sm.dq[tid], md.q_lower[tid], md.q_upper[tid], md.vel_limit[tid], md.effort[tid], sm.tau[tid], &act);
			sm.v1[tid] = act ? 1.0 : 0.0;
		}
		SYNC();
		double nact = 0.0;
		for (int j = 0; j < n; j++) nact += sm.v1[j];
		if (nact > 0.0) {
			for (int i = tid; i < n; i += T) {
				double sacc = 0.0;
				for (int j = 0; j < n; j++) sacc = fma(sm.Minv[i * LD + j], sm.tau[j], sacc);
				sm.v2[i] = (sm.v1[i] != 0.0) ? sacc : 0.0;
			}
			for (int e = tid; e < n * n; e += T) {
				int i = divn(e), j = e - i * n;
				sm.W[0][i * LD + j] = (sm.v1[i] != 0.0 && sm.v1[j] != 0.0) ? sm.Minv[i * LD + j] : ((i == j) ? 1.0 : 0.0);
			}
			SYNC();
			spd_inverse<T>(sm.W[0], sm.W[1], sm.W[2], sm.W[3], sm.dv, n, LD, tid);
			if (tid < n) {
				double y = 0.0;
				for (int j = 0; j < n; j++) y = fma(sm.W[1][tid * LD + j], sm.v2[j], y);
				double tv = sm.tau[tid];
				if (sm.v1[tid] != 0.0) tv = sm.v0[tid] + tv - y;
				if (P.torque_sat) {
					const double lim = md.effort[tid];
					tv = tv > lim ? lim : (tv < -lim ? -lim : tv);
				}
				sm.v3[tid] = tv;
			}
			SYNC();
			if (tid < n) sm.tau[tid] = sm.v3[tid];
		}
	}
	if (tid < n) {
		double tv = sm.tau[tid];
		if (P.gravity_comp) {
			// jointGravityVector: tau_g[j] = -(d c_com / d q_j) . (m g) summed over the subtree = -S_j . (composite gravity wrench)
			const double* Ic = sm.comp + 10 * tid;
			const double* z = sm.zax + 3 * tid;
			const double* o = fk + 12 * tid + 9;
			const double* g = md.gravity;
			double tg;
			if (md.jtype[tid] == 1) {
				double ax = Ic[1] - Ic[0] * o[0], ay = Ic[2] - Ic[0] * o[1], az = Ic[3] - Ic[0] * o[2];  // m (c - o)
				double cx = z[1] * az - z[2] * ay, cy = z[2] * ax - z[0] * az, cz = z[0] * ay - z[1] * ax;
				tg = -(cx * g[0] + cy * g[1] + cz * g[2]);
			} else {
				tg = -Ic[0] * (z[0] * g[0] + z[1] * g[1] + z[2] * g[2]);
			}
			tv += tg;
		}
		// a flagged instance (status 1) holds its last valid torques unless NaN was asked for (saip_batch_set_flagged_torque_policy)
		if (!singular) P.tau[(size_t)tid * P.ld + b] = tv;
		else if (P.flag_nan) P.tau[(size_t)tid * P.ld + b] = __builtin_nan("");
	}
	if (tid == 0) P.status[b] = singular ? 1 : ((truncated ? 2 : 0) | (sm.flag2 ? 4 : 0) | (blended ? 8 : 0));
	if (!singular) commit_integrators();  // a flagged instance advances nothing, like in the lane / eight-lane kernels
}

}  // namespace saip
