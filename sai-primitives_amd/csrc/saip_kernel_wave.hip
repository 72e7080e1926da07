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

#include "saip_device.h"
#include "saip_fk.h"
#include "saip_law.h"
#include "saip_wave_prims.h"

namespace saip {

struct WaveSmem {
	static constexpr int LDI = 33;
	double img[32 * LDI];  // staging for the kinematics scans, then the image of U^T Jp (joint task), read back in both operand forms
	double jd[32][12];     // per joint, about the world origin: motion axis S = (w, v), F = Ic S = (f, nO)
	double q[32], dq[32], tau[32], va[32], vb[32], vc[32], vd[32], act[32], diagM[32];
	double goal[96];
	double vw[8], Fum[8], Ff[8], xp[4], Rc[12];
	double f8[16], ff8[16], g8[16];
	int idx[16];
	int flag2;
};

__device__ __forceinline__ double wv_dot6(const double* S, const double* F) {  // S = (w, v), F = (f, nO): w . nO + v . f
	return fma(S[5], F[2], fma(S[4], F[1], fma(S[3], F[0], fma(S[2], F[5], fma(S[1], F[4], S[0] * F[3])))));
}

__global__ void __launch_bounds__(64, 2) saip_cycle_wave(const CycleParams P) {
	__shared__ WaveSmem sm;
	const int b = blockIdx.x;
	if (b >= P.B) return;
	const int lane = threadIdx.x, lo = lane & 15, hi = lane >> 4;
	const int j = lane & 31;  // joint owned in the per-joint phases (both halves of the wavefront do the same work)
	const int n = P.n, ld = P.ld;
	const ModelDev& md = *P.model;
	const bool jv = j < n;
	constexpr int LDI = WaveSmem::LDI;

	bool flagged = false;  // wave-uniform
	// ---------------------------------------------------------------- state
	const double qj = jv ? P.q[(size_t)j * ld + b] : 0.0;
	const double dqj = jv ? P.dq[(size_t)j * ld + b] : 0.0;
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
		sincos_joint(qj, &s, &c);
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
	// log-step prefix product over the chain: T_j <- T_{j-d} o T_j, (Ra, oa) o (Rb, ob) = (Ra Rb, oa + Ra ob)
	UNR for (int d = 1; d < 32; d <<= 1) {
		if (d < n) {
			if (lane < 32) {
				UNR for (int e = 0; e < 12; e++) sm.img[12 * j + e] = X[e];
			}
			wv_sync();
			if (j >= d) {
				double Xa[12], Xn[12];
				UNR for (int e = 0; e < 12; e++) Xa[e] = sm.img[12 * (j - d) + e];
				UNR for (int r = 0; r < 3; r++) {
					UNR for (int cc = 0; cc < 3; cc++) Xn[3 * r + cc] = Xa[3 * r] * X[cc] + Xa[3 * r + 1] * X[3 + cc] + Xa[3 * r + 2] * X[6 + cc];
					Xn[9 + r] = Xa[9 + r] + Xa[3 * r] * X[9] + Xa[3 * r + 1] * X[10] + Xa[3 * r + 2] * X[11];
				}
				UNR for (int e = 0; e < 12; e++) X[e] = Xn[e];
			}
			wv_sync();
		}
	}
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
	// composite inertias: log-step suffix sums (plain sums: everything is expressed about the world origin)
	UNR for (int d = 1; d < 32; d <<= 1) {
		if (d < n) {
			if (lane < 32) {
				UNR for (int e = 0; e < 10; e++) sm.img[10 * j + e] = Bd[e];
			}
			wv_sync();
			if (j + d < 32) {
				UNR for (int e = 0; e < 10; e++) Bd[e] += sm.img[10 * (j + d) + e];
			}
			wv_sync();
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
		}
	}
	wv_sync();
	// ---------------------------------------------------------------- M(q) in stored form: M_ij = S_min(i,j) . F_max(i,j); identity beyond n
	double Minv[8][2];
	{
		double Sc[2][6], Fc[2][6];
		UNR for (int t = 0; t < 2; t++)
			UNR for (int e = 0; e < 6; e++) {
				Sc[t][e] = sm.jd[16 * t + lo][e];
				Fc[t][e] = sm.jd[16 * t + lo][6 + e];
			}
		UNR for (int ks = 0; ks < 8; ks++) {
			const int i = 4 * ks + hi;
			double Si[6], Fi[6];
			UNR for (int e = 0; e < 6; e++) {
				Si[e] = sm.jd[i][e];
				Fi[e] = sm.jd[i][6 + e];
			}
			UNR for (int t = 0; t < 2; t++) {
				const int jc = 16 * t + lo;
				const double d1 = wv_dot6(Si, Fc[t]), d2 = wv_dot6(Sc[t], Fi);
				double val = (i <= jc) ? d1 : d2;
				if (i >= n || jc >= n) val = (i == jc) ? 1.0 : 0.0;
				Minv[ks][t] = val;
				if (i == jc) sm.diagM[i] = val;
			}
		}
	}
	{
		const double mp = wv_sweep<2>(Minv, (n + 3) >> 2, lo, hi);
		if (!(mp > 0.0)) flagged = true;
		wv_negate(Minv);
	}
	wv_sync();
	// ---------------------------------------------------------------- bounded inertia estimates: clamped entries, K^-1
	double Em[8][1], Kinv[2][1];
	int nclamp = 0;
	UNR for (int ks = 0; ks < 8; ks++) Em[ks][0] = 0.0;
	Kinv[0][0] = Kinv[1][0] = 0.0;
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
			const int myidx = lo < nclamp ? sm.idx[lo] : -1;  // column lo of E selects joint idx[lo]
			UNR for (int ks = 0; ks < 8; ks++) Em[ks][0] = (myidx == 4 * ks + hi) ? 1.0 : 0.0;
			double ME[8][1], K4[4][1];
			wv_mm<false, false>(Minv, Em, ME);  // M^-1 E (n x c)
			wv_mm<false, false>(Em, ME, K4);    // E^T M^-1 E (c x c)
			UNR for (int ks = 0; ks < 4; ks++) {
				const int i = 4 * ks + hi;
				if (i == lo) {
					if (i < nclamp) K4[ks][0] += 1.0 / (thr - sm.diagM[myidx]);
					else if (i < 8) K4[ks][0] = 1.0;
				}
			}
			const double mp = wv_sweep<1>(K4, 2, lo, hi);
			if (!(mp > 0.0)) flagged = true;
			Kinv[0][0] = -K4[0][0];
			Kinv[1][0] = -K4[1][0];
		}
	}
	// ---------------------------------------------------------------- task hierarchy
	double Np[8][2];
	UNR for (int ks = 0; ks < 8; ks++)
		UNR for (int t = 0; t < 2; t++) Np[ks][t] = (4 * ks + hi == 16 * t + lo && 4 * ks + hi < n) ? 1.0 : 0.0;
	unsigned commit_lo = 0, commit_hi = 0;
	for (int t = 0; t < P.nt && !flagged; t++) {
		const TaskDev& tk = P.tasks[t];
		const bool need_N = t + 1 < P.nt;
		const bool track = tk.has_ki || P.integ_always;
		for (int e = lane; e < tk.goal_comps; e += 64) sm.goal[e] = tk.goal[(size_t)e * ld + b];
		if (tk.decoupling == DEC_BIE && tk.bie_threshold != P.bie_thr) {
			flagged = true;
			break;
		}
		if (tk.type == TASK_MOTION_FORCE) {
			const int k = tk.k, bd = tk.body;
			// control point and frame of the task: the world frame of body `bd` sits in lane bd
			double Rb[9], ob[3], pw[3], Rcf[9];
			UNR for (int e = 0; e < 9; e++) Rb[e] = wv_readlane(X[e], bd);
			UNR for (int e = 0; e < 3; e++) ob[e] = wv_readlane(X[9 + e], bd);
			oct_mat3_vec(Rb, tk.pos, pw);
			UNR for (int e = 0; e < 3; e++) pw[e] += ob[e];
			UNR for (int r = 0; r < 3; r++)
				UNR for (int cc = 0; cc < 3; cc++) Rcf[3 * r + cc] = Rb[3 * r] * tk.rot[cc] + Rb[3 * r + 1] * tk.rot[3 + cc] + Rb[3 * r + 2] * tk.rot[6 + cc];
			if (lane == 0) {
				UNR for (int e = 0; e < 3; e++) sm.xp[e] = pw[e];
				UNR for (int e = 0; e < 9; e++) sm.Rc[e] = Rcf[e];
			}
			// Jb^T = Jw^T Bm (n x k) in stored form: lane (lo, hi) owns task direction lo of joints 4 ks + hi; column j of
			// JWorldFrame is (w_j x p + v_j; w_j) for the joints up to the body (MotionForceTask.cpp:262)
			double bmc[6];
			UNR for (int i = 0; i < 6; i++) bmc[i] = (lo < k) ? tk.Bm[i * 6 + (lo < 6 ? lo : 0)] : 0.0;
			double Jbt[8][1], vwp[6] = {0, 0, 0, 0, 0, 0};
			UNR for (int ks = 0; ks < 8; ks++) {
				const int jj = 4 * ks + hi;
				const double* S = sm.jd[jj];
				const bool on = jj <= bd;
				const double w0 = on ? S[0] : 0.0, w1 = on ? S[1] : 0.0, w2 = on ? S[2] : 0.0;
				const double c0 = on ? (w1 * pw[2] - w2 * pw[1] + S[3]) : 0.0, c1 = on ? (w2 * pw[0] - w0 * pw[2] + S[4]) : 0.0,
							 c2 = on ? (w0 * pw[1] - w1 * pw[0] + S[5]) : 0.0;
				const double dqv = sm.dq[jj];
				vwp[0] = fma(c0, dqv, vwp[0]);
				vwp[1] = fma(c1, dqv, vwp[1]);
				vwp[2] = fma(c2, dqv, vwp[2]);
				vwp[3] = fma(w0, dqv, vwp[3]);
				vwp[4] = fma(w1, dqv, vwp[4]);
				vwp[5] = fma(w2, dqv, vwp[5]);
				Jbt[ks][0] = fma(w2, bmc[5], fma(w1, bmc[4], fma(w0, bmc[3], fma(c2, bmc[2], fma(c1, bmc[1], c0 * bmc[0])))));
			}
			// task-frame velocities use the unprojected Jacobian (MotionForceTask.cpp:293-298): vw = Jw dq
			UNR for (int e = 0; e < 6; e++) {
				const double s = wv_hsum(vwp[e]);
				if (lane == 0) sm.vw[e] = s;
			}
			double Jh[2][2], Jht[8][1];
			wv_mm<false, false>(Jbt, Np, Jh);   // Jh = Jb N_prec (k x n)
			wv_mm<false, false>(Np, Jbt, Jht);  // Jh^T (n x k)
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
				const double mp = wv_sweep<1>(Gc, 2, lo, hi);
				certified = (mp > 1e-13) && ok_s0 && (tr > 0.0) && !nonfinite;
			}
			if (!certified) {  // outside the non-singular branch, or too close to call: the general kernel decides (exact eigen-solve)
				flagged = true;
				break;
			}
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
			if (!pad_sweep(A2, Lam)) {
				flagged = true;
				break;
			}
			if (tk.decoupling == DEC_FULL) {
				UNR for (int ks = 0; ks < 4; ks++) Lmod[ks][0] = Lam[ks][0];
			} else if (tk.decoupling == DEC_IMPEDANCE) {
				UNR for (int ks = 0; ks < 4; ks++) Lmod[ks][0] = (4 * ks + hi == lo && lo < 8) ? 1.0 : 0.0;
			} else if (nclamp == 0) {
				UNR for (int ks = 0; ks < 4; ks++) Lmod[ks][0] = Lam[ks][0];
			} else {
				double Ytt[2][1], Zk[2][1], AB[2][1];
				wv_mm<false, false>(Em, T1t, Ytt);    // (T1 E)^T  (c x k)
				wv_mm<false, false>(Kinv, Ytt, Zk);   // K^-1 (T1 E)^T
				AB[0][0] = A2[0][0];
				AB[1][0] = A2[1][0];
				wv_mm<true, true>(Ytt, Zk, AB);       // A - (T1 E) K^-1 (T1 E)^T
				if (!pad_sweep(AB, Lmod)) {
					flagged = true;
					break;
				}
			}
			// control law on one lane (MotionForceTask.cpp:286-509), then g = Lambda_mod B^T F_um + B^T F_f (SingularityHandler.cpp:307-309)
			wv_sync();
			if (lane == 0) {
				double ip[3] = {0, 0, 0}, io[3] = {0, 0, 0}, ifm[6] = {0, 0, 0, 0, 0, 0};
				if (track)
					for (int i = 0; i < 3; i++) {
						ip[i] = tk.integ[(size_t)i * ld + b];
						io[i] = tk.integ[(size_t)(3 + i) * ld + b];
					}
				const bool closed = tk.cl_force || tk.cl_moment;
				if (closed)
					for (int i = 0; i < 6; i++) ifm[i] = tk.integ[(size_t)(6 + i) * ld + b];
				bool ovf = false;
				law_motion_force_general<true>(&tk, sm.vw, sm.xp, sm.Rc, sm.goal, ip, io, ifm, sm.Fum, sm.Ff, tk.popc ? tk.popc + b : nullptr, ld, &ovf);
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
			if (lane < 16) {
				double s = 0.0, s2 = 0.0;
				UNR for (int i = 0; i < 6; i++) {
					s = fma(bmc[i], sm.Fum[i], s);
					s2 = fma(bmc[i], sm.Ff[i], s2);
				}
				sm.f8[lane] = s;  // bmc is zero for lo >= k
				sm.ff8[lane] = s2;
			}
			wv_sync();
			{
				double Lm2[2][1] = {{Lmod[0][0]}, {Lmod[1][0]}}, fr[2], y[1];
				wv_load_rows(sm.f8, hi, fr);
				wv_matvec(Lm2, fr, y);
				if (lane < 16) sm.g8[lane] = y[0] + sm.ff8[lane];
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
			if (need_N) {  // N_prec <- N N_prec, N^T = I - Jh^T (Lambda T1)
				double Lm2[2][1] = {{Lam[0][0]}, {Lam[1][0]}}, Jbart[2][2], Nt[8][2], Np2[8][2];
				wv_mm<false, false>(Lm2, T1, Jbart);
				UNR for (int ks = 0; ks < 8; ks++)
					UNR for (int tt = 0; tt < 2; tt++) Nt[ks][tt] = (4 * ks + hi == 16 * tt + lo) ? 1.0 : 0.0;
				wv_mm<true, true>(Jh, Jbart, Nt);
				wv_mm<false, false>(Nt, Np, Np2);
				UNR for (int ks = 0; ks < 8; ks++)
					UNR for (int tt = 0; tt < 2; tt++) Np[ks][tt] = Np2[ks][tt];
			}
			wv_sync();
		} else {
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
						cur = qj;
						vel = dqj;
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
					double ie = track ? tk.integ[(size_t)j * ld + b] : 0.0;
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
			// matrixRangeBasis(Jp) (JointTask.cpp:233): column-pivoted Gram-Schmidt, accepted only on a clean rank gap (next pivot below
			// 1e-10 or above 1e-2 of the first); row s of U^T Jp goes to the LDS image, U^T xa / U^T f are taken on the fly
			double xar[8], xbr[8];
			wv_load_rows(sm.vc, hi, xar);
			wv_load_rows(sm.vb, hi, xbr);
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
				if (s == 0) {
					const double frob2 = wv_rowsum(nr[0] + nr[1]);
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
					pa = fma(u[ks], xar[ks], pa);
					pb = fma(u[ks], xbr[ks], pb);
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
				break;
			}
			const int r = rank;
			wv_sync();
			if (r > 0) {
				double Jh[8][2], Jht[8][2];
				UNR for (int ks = 0; ks < 8; ks++)
					UNR for (int tt = 0; tt < 2; tt++) {
						const int rowi = 4 * ks + hi, coli = 16 * tt + lo;
						Jh[ks][tt] = rowi < r ? sm.img[rowi * LDI + coli] : 0.0;   // U^T Jp (r x n)
						Jht[ks][tt] = coli < r ? sm.img[coli * LDI + rowi] : 0.0;  // its transpose (n x r)
					}
				const int nb = (r + 3) >> 2;
				double T1t[8][2], A[8][2];
				wv_mm<false, false>(Minv, Jht, T1t);  // M^-1 Jh^T (n x r)
				wv_mm<false, false>(T1t, Jht, A);     // Jh M^-1 Jh^T
				// Lambda_mod (JointTask.cpp:253-267) first: its action on b1 = U^T f, then Lambda itself in place
				double ymod[2];
				{
					double LM[8][2];
					UNR for (int ks = 0; ks < 8; ks++)
						UNR for (int tt = 0; tt < 2; tt++) LM[ks][tt] = A[ks][tt];
					bool ident = false, same = false;
					if (tk.decoupling == DEC_FULL || (tk.decoupling == DEC_BIE && nclamp == 0)) same = true;
					else if (tk.decoupling == DEC_IMPEDANCE) ident = true;
					else {
						double Ytt[2][2], Zk[2][2];
						wv_mm<false, false>(Em, T1t, Ytt);   // (T1 E)^T (c x r)
						wv_mm<false, false>(Kinv, Ytt, Zk);  // K^-1 (T1 E)^T
						wv_mm<true, true>(Ytt, Zk, LM);      // A - (T1 E) K^-1 (T1 E)^T
					}
					double b1r[8];
					wv_load_rows(sm.act, hi, b1r);
					if (ident) {
						ymod[0] = sm.act[lo];
						ymod[1] = sm.act[16 + lo];
					} else {
						if (!same) {
							wv_pad_identity(LM, r, 4 * nb, lo, hi);
							const double mp = wv_sweep<2>(LM, nb, lo, hi);
							if (!(mp > 0.0)) flagged = true;
							wv_negate(LM);
							wv_matvec(LM, b1r, ymod);
						}
					}
					wv_pad_identity(A, r, 4 * nb, lo, hi);
					const double mp = wv_sweep<2>(A, nb, lo, hi);
					if (!(mp > 0.0)) flagged = true;
					wv_negate(A);  // Lambda
					if (same) wv_matvec(A, b1r, ymod);
				}
				if (flagged) break;
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
				{
					double gr[8], y[2];
					wv_load_rows(sm.vd, hi, gr);
					wv_matvec(Jh, gr, y);  // tau += Jh^T g
					if (hi == 0) {
						sm.tau[lo] += y[0];
						sm.tau[16 + lo] += y[1];
					}
				}
				if (need_N) {
					double T1[8][2], Jbart[8][2], Nt[8][2], Np2[8][2];
					wv_mm<false, false>(Jht, Minv, T1);   // Jh M^-1 (r x n)
					wv_mm<false, false>(A, T1, Jbart);    // Lambda T1
					UNR for (int ks = 0; ks < 8; ks++)
						UNR for (int tt = 0; tt < 2; tt++) {
							Nt[ks][tt] = (4 * ks + hi == 16 * tt + lo) ? 1.0 : 0.0;
							if (4 * ks + hi >= r) Jbart[ks][tt] = 0.0;  // the identity padding of Lambda is not part of it
						}
					wv_mm<true, true>(Jh, Jbart, Nt);
					wv_mm<false, false>(Nt, Np, Np2);
					UNR for (int ks = 0; ks < 8; ks++)
						UNR for (int tt = 0; tt < 2; tt++) Np[ks][tt] = Np2[ks][tt];
				}
			}
			// r == 0: no controllable direction left, N = I (JointTask.cpp:234-239)
			wv_sync();
		}
	}
	wv_sync();
	// ---------------------------------------------------------------- post-processing, RobotController.cpp:86-116
	double tv = sm.tau[j];
	if (P.torque_sat && jv) {
		const double lim = md.effort[j];
		tv = tv > lim ? lim : (tv < -lim ? -lim : tv);
	}
	if (P.jla && !flagged) {
		// tau = JLA.computeTorques(tau) + N_constraints^T tau,  N_constraints^T = I - S^T (S M^-1 S^T)^-1 S M^-1 (padded to n x n)
		bool actj = false;
		double jl = 0.0;
		if (jv) jl = jla_joint(qj, dqj, md.q_lower[j], md.q_upper[j], md.vel_limit[j], md.effort[j], tv, &actj);
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
	if (P.gravity_comp) tv += tg;
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
	if (P.flag_list && b == 0 && lane == 0) *P.flag_count_next = 0;  // the other list: its last readers finished a cycle ago
	if (!flagged) {  // integrators: commit what the control laws advanced them to (a flagged instance advances nothing)
		for (int t = 0; t < P.nt; t++) {
			const TaskDev& tk = P.tasks[t];
			const int rows = tk.type == TASK_MOTION_FORCE ? 6 : tk.m;
			if ((commit_lo >> t) & 1u)
				for (int i = lane; i < rows; i += 64) tk.integ[(size_t)i * ld + b] = tk.integ_new[(size_t)i * ld + b];
			if ((commit_hi >> t) & 1u)
				for (int i = lane; i < 6; i += 64) tk.integ[(size_t)(6 + i) * ld + b] = tk.integ_new[(size_t)(6 + i) * ld + b];
		}
	}
}

hipError_t launch_cycle_wave(const CycleParams& P, hipStream_t stream) {
	hipLaunchKernelGGL(saip_cycle_wave, dim3(P.B), dim3(64), 0, stream, P);
	return hipGetLastError();
}

}  // namespace saip
