// Throughput control-cycle kernel: ONE LANE PER ROBOT INSTANCE, the whole cycle in registers (dof N in {6,7,8}).
//
// Every matrix of an instance lives in the lane's VGPRs (fully unrolled, compile-time indices only); 64 instances
// per wavefront run in lock-step with zero cross-lane traffic, so the FP64 vector pipe is the only shared resource.
// State/goals are SoA in HBM ([component][instance]) => every global access is a 512-byte coalesced row per wave.
// Cold per-lane data (joint axes/origins, control frames, sin/cos) is parked in LDS as [element][lane] (conflict-free).
// Template parameters: N = dof, RJ = static bound on the rank of the joint tasks (host picks the smallest instantiated
// value >= n - sum of the motion-force ranks above); motion-force tasks always use 6 rows (rows >= k are zero padding).
//
// Same arithmetic contract as the general kernel (saip_kernel_wg.hip) and the reference lines cited there, with two
// cheaper replacements for the reference's SVDs on the common, well-conditioned case:
//   * SingularityHandler's branch predicate (sigma_i/sigma_0 >= s_max for all i, sigma_0 >= 1e-3) is CERTIFIED without an
//     eigen-solve: lambda_max(G) <= u := ||G^4||_F^(1/4) (trace-power bound, two squarings of the 6 x 6 Gram matrix G),
//     and G - s_max^2 u I > 0 is checked by an LDL^T sweep.  Both hold => fully non-singular branch, rigorously.
//   * matrixRangeBasis(S N_prec) is an orthonormal basis from a column-pivoted Gram-Schmidt sweep, accepted only when the
//     rank gap is clean (next pivot below 1e-10 of the first, or above 1e-2).
// Lanes that cannot be certified (close to a threshold) take the EXACT path inside this same kernel: the wavefront
// cooperates on that lane's small Gram matrix in LDS with the parallel Jacobi eigen-solver of the general kernel
// (wave-uniform branch on a ballot: no cost when no lane needs it, no CPU fallback ever).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>

#include "saip_device.h"
#include "saip_law.h"
#include "saip_wg_linalg.h"
#include "saip_fk.h"

namespace saip {

#define UNR _Pragma("unroll")
#define CONSTANT_AS __attribute__((address_space(4)))
#if defined(SAIP_STAMP)
// diagnostic build only (never shipped): s_memtime stamps of block 0 / lane 0 at every phase boundary
__device__ unsigned long long saip_stamps[256];
#define PHASE_FENCE()                                                                      \
	do {                                                                                   \
		__builtin_amdgcn_sched_barrier(0);                                                 \
		if (blockIdx.x == 0 && threadIdx.x == 0) {                                         \
			saip_stamps[2 * __COUNTER__] = __builtin_amdgcn_s_memtime();                   \
			saip_stamps[2 * (__COUNTER__ - 1) + 1] = __LINE__;                             \
		}                                                                                  \
		__builtin_amdgcn_sched_barrier(0);                                                 \
	} while (0)
#define DIAG_STAMP() PHASE_FENCE()
#elif defined(SAIP_FENCE)
#define PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
// Scheduling barriers between the phases paid off under the default machine scheduler; with the max-ilp strategy the kernels are
// 0.4-1.5 % faster without them (A/B in one session, tools/ab_multi.sh), so they are now a diagnostic option (-DSAIP_FENCE).
#define PHASE_FENCE()
#endif
#ifndef DIAG_STAMP
#define DIAG_STAMP()  // extra stamps of the diagnostic build only: no scheduling effect in the shipped kernels
#endif

typedef const ModelDev CONSTANT_AS* ModelCPtr;
typedef const TaskDev CONSTANT_AS* TaskCPtr;

// Makes a wave-uniform pointer opaque to the optimiser at this point: constant loads through the result cannot be hoisted
// above it (they would otherwise all be issued at kernel entry and exhaust the SGPR file -> v_writelane/v_readlane spills).
template <typename Ptr>
__device__ __forceinline__ Ptr launder(Ptr p) {
#ifndef SAIP_NO_LAUNDER
	asm volatile("" : "+s"(p));
#endif
	return p;
}

template <int N, bool PF, bool LEAN = false>
struct LaneSmem {
	static constexpr int LD = 9;
	static constexpr int PFN = PF ? (2 * N + 24 + 3 * N) : 1;  // q, dq, motion-force goal block, joint goal block (SIG_MF_JT)
	double pf[PFN][64];
	static constexpr int PARK = 6 * N + 24 + (LEAN ? 0 : 10 * N);  // joint axes + origins, two control frames, (per-body spatial inertias)
	double park[PARK][64];
	static constexpr bool MINV_LDS = !LEAN;              // latency variant: M^-1 and M_BIE^-1 parked in LDS, rows streamed on use
	double minv[MINV_LDS ? N * (N + 1) : 1][64];        // two symmetric matrices, N(N+1)/2 rows each
	double qsh[N][64];                                   // joint angles, fetched with one exposed memory latency for the rolled FK loop
	double dqsh[(PF || LEAN) ? 1 : N][64];               // joint velocities, fetched in the same burst (latency variant)
	static constexpr bool DQ_LDS = !(PF || LEAN);
	double A[8 * LD], V[8 * LD], Jp[8 * LD], cs[4 * 5];
	int colidx[8];
	int flag, rank;
};

// Bounded inertia estimate as a rank-one correction.  M_BIE = M + sum_c delta_c e_c e_c^T with delta_c = max(0, thr - M_cc).
// When at most one joint of the instance is clamped (the usual case: a light wrist joint),
//   M_BIE^-1 = M^-1 - coef v v^T,  v = sum_c delta_c M^-1[:,c] (= delta m_c),  coef = 1 / (delta (1 + delta M^-1_cc)),
// so Jh M_BIE^-1 Jh^T = A - coef t t^T with t = Jh v, and Lambda_mod follows from Lambda by Sherman-Morrison: no second
// Cholesky of M and no second k x k inverse.  use_sm is wave-uniform; wavefronts holding an instance with two or more
// clamped joints take the full path (M_BIE^-1 by Cholesky) for all their lanes.
template <int N>
struct BieRankOne {
	double v[N];
	double coef;
	bool use_sm;
};

// Symmetric N x N matrix that lives either in registers or parked in LDS as N(N+1)/2 lane-interleaved rows (conflict-free).
// row() streams one row into registers: consumers are written row-major so that an LDS-resident matrix costs N reads per N*R FMAs.
template <int N, bool INLDS>
struct SymMat {
	const double (*reg)[N];
	const double (*lds)[64];
	int lane;
	__device__ __forceinline__ void row(const int l, double (&r)[N]) const {
		UNR for (int j = 0; j < N; j++) {
			if (INLDS) {
				const int a = l > j ? l : j, b = l > j ? j : l;
				r[j] = lds[a * (a + 1) / 2 + b][lane];
			} else {
				r[j] = reg[l][j];
			}
		}
	}
};

// Ainv = A^-1 for a symmetric positive definite R x R register matrix (lower triangle of A is read)
template <int R>
__device__ __forceinline__ void spd_inv_reg(const double (&A)[R][R], double (&Ainv)[R][R]) {
	double L[R][R], X[R][R], dinv[R];
	UNR for (int i = 0; i < R; i++)
		UNR for (int j = 0; j <= i; j++) L[i][j] = A[i][j];
	UNR for (int k = 0; k < R; k++) {
		const double rs = rsqrt(L[k][k]);
		dinv[k] = rs;  // 1 / L_kk
		UNR for (int i = k + 1; i < R; i++) L[i][k] *= rs;
		UNR for (int j = k + 1; j < R; j++)
			UNR for (int i = j; i < R; i++) L[i][j] = fma(-L[i][k], L[j][k], L[i][j]);
	}
	UNR for (int c = 0; c < R; c++) {  // X = L^-1 (lower)
		X[c][c] = dinv[c];
		UNR for (int i = c + 1; i < R; i++) {
			double s = 0.0;
			UNR for (int k = c; k < i; k++) s = fma(L[i][k], X[k][c], s);
			X[i][c] = -s * dinv[i];
		}
	}
	UNR for (int i = 0; i < R; i++)
		UNR for (int j = 0; j <= i; j++) {
			double s = 0.0;
			UNR for (int k = i; k < R; k++) s = fma(X[k][i], X[k][j], s);
			Ainv[i][j] = s;
			Ainv[j][i] = s;
		}
}

// operationalSpaceMatrices in the range basis.  Jh: R x N (rows >= r_lane are zero).  Outputs Lambda, Lambda_mod, T1 = Jh Minv.
template <int N, int R, bool INLDS>
__device__ __forceinline__ void opspace_lambda(const double (&Jh)[R][N], const SymMat<N, INLDS>& Minv, const SymMat<N, INLDS>& MBinv,
											   const BieRankOne<N>& bie, const int r_lane, const int decoupling, double (&Lam)[R][R], double (&Lmod)[R][R],
											   double (&T1)[R][N]) {
	double A[R][R];
	UNR for (int a = 0; a < R; a++)
		UNR for (int j = 0; j < N; j++) T1[a][j] = 0.0;
	UNR for (int l = 0; l < N; l++) {
		double mr[N];
		Minv.row(l, mr);
		UNR for (int a = 0; a < R; a++)
			UNR for (int j = 0; j < N; j++) T1[a][j] = fma(Jh[a][l], mr[j], T1[a][j]);
	}
	UNR for (int a = 0; a < R; a++)
		UNR for (int c = 0; c <= a; c++) {
			double s = 0.0;
			UNR for (int j = 0; j < N; j++) s = fma(T1[a][j], Jh[c][j], s);
			if (a == c) s = (a >= r_lane) ? 1.0 : s;  // zero rows beyond the lane's rank: pad so the factorisation stays regular
			A[a][c] = s;
		}
	spd_inv_reg<R>(A, Lam);
	if (decoupling == DEC_FULL) {
		UNR for (int a = 0; a < R; a++)
			UNR for (int c = 0; c < R; c++) Lmod[a][c] = Lam[a][c];
	} else if (decoupling == DEC_IMPEDANCE) {
		UNR for (int a = 0; a < R; a++)
			UNR for (int c = 0; c < R; c++) Lmod[a][c] = (a == c) ? 1.0 : 0.0;
	} else if (bie.use_sm) {
		// Lambda_mod = (A - coef t t^T)^-1 = Lambda + gamma (Lambda t)(Lambda t)^T,  gamma = coef / (1 - coef t^T Lambda t)
		double t[R], lt[R], q = 0.0;
		UNR for (int a = 0; a < R; a++) {
			double sacc = 0.0;
			UNR for (int j = 0; j < N; j++) sacc = fma(Jh[a][j], bie.v[j], sacc);
			t[a] = sacc;
		}
		UNR for (int a = 0; a < R; a++) {
			double sacc = 0.0;
			UNR for (int c = 0; c < R; c++) sacc = fma(Lam[a][c], t[c], sacc);
			lt[a] = sacc;
			q = fma(t[a], sacc, q);
		}
		const double gamma = bie.coef / (1.0 - bie.coef * q);
		UNR for (int a = 0; a < R; a++)
			UNR for (int c = 0; c <= a; c++) {
				const double val = fma(gamma * lt[a], lt[c], Lam[a][c]);
				Lmod[a][c] = val;
				Lmod[c][a] = val;
			}
	} else if (INLDS) {
		double Tb[R][N];
		UNR for (int a = 0; a < R; a++)
			UNR for (int j = 0; j < N; j++) Tb[a][j] = 0.0;
		UNR for (int l = 0; l < N; l++) {
			double mr[N];
			MBinv.row(l, mr);
			UNR for (int a = 0; a < R; a++)
				UNR for (int j = 0; j < N; j++) Tb[a][j] = fma(Jh[a][l], mr[j], Tb[a][j]);
		}
		UNR for (int a = 0; a < R; a++)
			UNR for (int c = 0; c <= a; c++) {
				double s = 0.0;
				UNR for (int j = 0; j < N; j++) s = fma(Tb[a][j], Jh[c][j], s);
				if (a == c) s = (a >= r_lane) ? 1.0 : s;
				A[a][c] = s;
			}
		spd_inv_reg<R>(A, Lmod);
	} else {
		UNR for (int a = 0; a < R; a++) {
			double tb[N];  // row a of Jh M_BIE^-1, consumed immediately
			UNR for (int j = 0; j < N; j++) {
				double s = 0.0;
				UNR for (int l = 0; l < N; l++) s = fma(Jh[a][l], MBinv.reg[l][j], s);
				tb[j] = s;
			}
			UNR for (int c = 0; c <= a; c++) {
				double s = 0.0;
				UNR for (int j = 0; j < N; j++) s = fma(tb[j], Jh[c][j], s);
				if (a == c) s = (a >= r_lane) ? 1.0 : s;
				A[a][c] = s;
			}
		}
		spd_inv_reg<R>(A, Lmod);
	}
}

// tau += Jh^T g;  N_prec <- (I - Minv Jh^T Lambda Jh) N_prec  (skipped for the last task)
template <int N, int R>
__device__ __forceinline__ void apply_task(const double (&Jh)[R][N], const double (&g)[R], const double (&Lam)[R][R], const double (&T1)[R][N],
										   const int r_lane, const bool first, const bool last, double (&tau)[N], double (&Np)[N][N]) {
	UNR for (int j = 0; j < N; j++) {
		double s = 0.0;
		UNR for (int a = 0; a < R; a++) s = fma(Jh[a][j], g[a], s);
		tau[j] += s;
	}
	if (last) return;
	// Jbar^T = Lambda T1 (R x N) is consumed one column at a time (never materialised: register pressure)
	if (first) {
		UNR for (int i = 0; i < N; i++) {
			double t2[R];
			UNR for (int a = 0; a < R; a++) {
				double s = 0.0;
				UNR for (int c = 0; c < R; c++) s = fma(Lam[a][c], T1[c][i], s);
				t2[a] = (a < r_lane) ? s : 0.0;
			}
			UNR for (int j = 0; j < N; j++) {
				double s = (i == j) ? 1.0 : 0.0;
				UNR for (int a = 0; a < R; a++) s = fma(-t2[a], Jh[a][j], s);
				Np[i][j] = s;
			}
		}
	} else {
		// N N_prec = N_prec - Jbar (Jh N_prec)
		double JN[R][N];
		UNR for (int a = 0; a < R; a++)
			UNR for (int j = 0; j < N; j++) {
				double s = 0.0;
				UNR for (int l = 0; l < N; l++) s = fma(Jh[a][l], Np[l][j], s);
				JN[a][j] = s;
			}
		UNR for (int i = 0; i < N; i++) {
			double t2[R];
			UNR for (int a = 0; a < R; a++) {
				double s = 0.0;
				UNR for (int c = 0; c < R; c++) s = fma(Lam[a][c], T1[c][i], s);
				t2[a] = (a < r_lane) ? s : 0.0;
			}
			UNR for (int j = 0; j < N; j++) {
				double s = Np[i][j];
				UNR for (int a = 0; a < R; a++) s = fma(-t2[a], JN[a][j], s);
				Np[i][j] = s;
			}
		}
	}
}

template <int N, bool PF, bool GL, typename SMT>
__device__ __forceinline__ void lane_motion_force_task(const CycleParams& P, SMT& sm, const ModelCPtr md, const TaskCPtr tasks, const int t, const int lane, const int b,
		const bool live, const bool first, const bool last, const SymMat<N, SMT::MINV_LDS>& Minv, const SymMat<N, SMT::MINV_LDS>& MBinv,
		const BieRankOne<N>& bie, double (&Np)[N][N], double (&tau)[N], bool& singular) {
	constexpr int LD = SMT::LD;
	constexpr int P_Z = 0, P_O = 3 * N, P_CAP = 6 * N;
	const size_t ld = P.ld;
	TaskCPtr tk = launder(tasks + t);
	const bool track = (tk->has_ki || P.integ_always);
	(void)LD; (void)P_Z; (void)P_O; (void)P_CAP; (void)md;
	// ======================================================== MotionForceTask (6 rows; rows >= k are zero)
	const int k = tk->k, bd = tk->body;
	const int cbase = P_CAP + ((t == P.mf_task[1]) ? 12 : 0);
	double pw[3];
	UNR for (int e = 0; e < 3; e++) pw[e] = sm.park[cbase + e][lane];
	double Jh[6][N], vw[6];
	UNR for (int e = 0; e < 6; e++) vw[e] = 0.0;
	{
		// world Jacobian [Jv; Jw] of the control point (JWorldFrame), task velocities, rows in the basis of range(P)
		double Jw[6][N];
		UNR for (int j = 0; j < N; j++) {
			double z[3], o[3];
			UNR for (int e = 0; e < 3; e++) {
				z[e] = sm.park[P_Z + 3 * j + e][lane];
				o[e] = sm.park[P_O + 3 * j + e][lane];
			}
			const bool on = (j <= bd);
			{  // branch-free: revolute (z x r; z) or prismatic (z; 0), zero beyond the task's body
				const bool rev = md->jtype[j] == 1;
				const double rx = pw[0] - o[0], ry = pw[1] - o[1], rz = pw[2] - o[2];
				const double c0 = z[1] * rz - z[2] * ry, c1 = z[2] * rx - z[0] * rz, c2 = z[0] * ry - z[1] * rx;
				Jw[0][j] = on ? (rev ? c0 : z[0]) : 0.0;
				Jw[1][j] = on ? (rev ? c1 : z[1]) : 0.0;
				Jw[2][j] = on ? (rev ? c2 : z[2]) : 0.0;
				Jw[3][j] = (on && rev) ? z[0] : 0.0;
				Jw[4][j] = (on && rev) ? z[1] : 0.0;
				Jw[5][j] = (on && rev) ? z[2] : 0.0;
			}
			const double dqj = PF ? sm.pf[N + j][lane] : (SMT::DQ_LDS ? sm.dqsh[j][lane] : P.dq[(size_t)j * ld + b]);
			UNR for (int e = 0; e < 6; e++) vw[e] = fma(Jw[e][j], dqj, vw[e]);
		}
		if (tk->bm_identity) {
			if (first) {
				UNR for (int a = 0; a < 6; a++)
					UNR for (int j = 0; j < N; j++) Jh[a][j] = Jw[a][j];
			} else {
				UNR for (int a = 0; a < 6; a++)
					UNR for (int j = 0; j < N; j++) {
						double s = 0.0;
						UNR for (int l = 0; l < N; l++) s = fma(Jw[a][l], Np[l][j], s);
						Jh[a][j] = s;
					}
			}
		} else {
			double Jb[6][N];
			UNR for (int a = 0; a < 6; a++)
				UNR for (int j = 0; j < N; j++) {
					double s = 0.0;
					UNR for (int e = 0; e < 6; e++) s = fma(tk->Bm[e * 6 + a], Jw[e][j], s);  // columns >= k of Bm are zero
					Jb[a][j] = s;
				}
			if (first) {
				UNR for (int a = 0; a < 6; a++)
					UNR for (int j = 0; j < N; j++) Jh[a][j] = Jb[a][j];
			} else {
				UNR for (int a = 0; a < 6; a++)
					UNR for (int j = 0; j < N; j++) {
						double s = 0.0;
						UNR for (int l = 0; l < N; l++) s = fma(Jb[a][l], Np[l][j], s);
						Jh[a][j] = s;
					}
			}
		}
	}
	PHASE_FENCE();
	// ---- SingularityHandler branch predicate on G = Jh Jh^T (k x k block; padded rows give zero rows/columns)
	bool need_exact = false;
	double G[6][6];
	{
		double tr = 0.0;
		UNR for (int a = 0; a < 6; a++)
			UNR for (int c = 0; c <= a; c++) {
				double s = 0.0;
				UNR for (int j = 0; j < N; j++) s = fma(Jh[a][j], Jh[c][j], s);
				G[a][c] = s;
				G[c][a] = s;
				if (a == c) tr += s;
			}
		const double itr = 1.0 / tr;
		const double smax2 = tk->s_max * tk->s_max, tol2 = tk->s_abs_tol * tk->s_abs_tol;
		// LDL^T pivots of Gs - shift I on the k x k block (padded diagonal entries are set to 1): all > 0 <=> lambda_min(Gs) > shift
		auto ldl_positive = [&](const double shift) {
			double Gs[6][6];
			UNR for (int a = 0; a < 6; a++)
				UNR for (int c = 0; c <= a; c++) Gs[a][c] = G[a][c] * itr;
			bool pos = true;
			UNR for (int a = 0; a < 6; a++) Gs[a][a] = (a < k) ? Gs[a][a] - shift : 1.0;
			UNR for (int kk = 0; kk < 6; kk++) {
				const double d = Gs[kk][kk];
				pos = pos && (d > 1e-13);
				const double id = 1.0 / d;
				UNR for (int i = kk + 1; i < 6; i++) {
					const double lik = Gs[i][kk] * id;
					UNR for (int j = kk + 1; j <= i; j++) Gs[i][j] = fma(-lik, Gs[j][kk], Gs[i][j]);
				}
			}
			return pos;
		};
		// stage 1: lambda_max(Gs) <= trace(Gs) = 1 and lambda_max(G) >= trace(G) / k.  Passes whenever the smallest eigenvalue is above
		// s_max^2 of the TRACE -- the usual case -- and costs one LDL^T sweep.
		// Tried for the first task of the stack only: a task projected through an earlier nullspace has a flatter, smaller spectrum
		// and mostly needs stage 2 anyway.
		const bool cheap = first && (tr > 0.0) && (tr >= (double)k * tol2) && ldl_positive(smax2);
		need_exact = !cheap;
		if (__any(need_exact)) {
			// stage 2 (some lane of the wavefront is closer to the bound): the tighter trace-power bound lambda_max <= ||Gs^4||_F^(1/4)
			double G2[6][6], Gt[6][6];
			UNR for (int a = 0; a < 6; a++)
				UNR for (int c = 0; c < 6; c++) G2[a][c] = G[a][c] * itr;
			UNR for (int rep = 0; rep < 2; rep++) {  // G2 <- G2^2: Gs^2, Gs^4
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
			double t8 = 0.0;  // trace(Gs^8) = ||Gs^4||_F^2
			UNR for (int a = 0; a < 6; a++)
				UNR for (int c = 0; c < 6; c++) t8 = fma(G2[a][c], G2[a][c], t8);
			const double u = sqrt(sqrt(sqrt(t8)));  // lambda_max(Gs) <= u <= k^(1/8) lambda_max(Gs)
			const bool ok_s0 = (u * tk->cert_kroot * tr >= tol2);
			const bool pos = ldl_positive(smax2 * u);
			need_exact = need_exact && !(ok_s0 && pos && tr > 0.0);
			if (__any(need_exact)) {
				// stage 3, the opposite certificate: with l = max_i Gs_ii <= lambda_max(Gs), a NEGATIVE LDL^T pivot of Gs - s_max^2 l I (all earlier
				// pivots positive: Sylvester) proves lambda_min < s_max^2 lambda_max -- outside the non-singular branch for sure; so does
				// trace(G) < s_abs_tol^2.  Without it every near-singular lane sent its whole wavefront through the eigen-solve just to confirm.
				double lmx = 0.0;
				UNR for (int a = 0; a < 6; a++) lmx = (a < k) ? fmax(lmx, G[a][a] * itr) : lmx;
				double Gs[6][6];
				UNR for (int a = 0; a < 6; a++)
					UNR for (int c = 0; c <= a; c++) Gs[a][c] = G[a][c] * itr;
				UNR for (int a = 0; a < 6; a++) Gs[a][a] = (a < k) ? Gs[a][a] - smax2 * lmx : 1.0;
				bool neg = false, clean = true;
				UNR for (int kk = 0; kk < 6; kk++) {
					const double dpv = Gs[kk][kk];
					neg = neg || (clean && dpv < -1e-12);
					clean = clean && (dpv > 1e-13);
					const double id = 1.0 / dpv;
					UNR for (int i = kk + 1; i < 6; i++) {
						const double lik = Gs[i][kk] * id;
						UNR for (int j = kk + 1; j <= i; j++) Gs[i][j] = fma(-lik, Gs[j][kk], Gs[i][j]);
					}
				}
				if (need_exact && ((tr > 0.0 && neg) || tr < tol2)) {
					singular = true;
					need_exact = false;
				}
			}
		}
	}
	// exact path (rare): wave-cooperative Jacobi eigen-solve of this lane's Gram matrix
	unsigned long long need = __ballot(need_exact);
	while (need) {
		const int src = __builtin_ctzll(need);
		need &= need - 1;
		if (lane == src) {
			UNR for (int a = 0; a < 6; a++)
				UNR for (int c = 0; c < 6; c++) sm.A[a * LD + c] = G[a][c];
		}
		__syncthreads();
		jacobi_eig<64>(sm.A, nullptr, k, LD, sm.cs, &sm.flag, lane);
		if (lane == src) {
			double lmax = 0.0;
			for (int i = 0; i < k; i++) lmax = fmax(lmax, sm.A[i * LD + i]);
			const double s0 = sqrt(lmax);
			bool sing = !(s0 >= tk->s_abs_tol);
			for (int i = 0; i < k; i++)
				if (sqrt(fmax(sm.A[i * LD + i], 0.0)) / s0 < tk->s_max) sing = true;
			if (sing) singular = true;
		}
		__syncthreads();
	}
	PHASE_FENCE();
	double Lam[6][6], Lmod[6][6], T1[6][N];
	opspace_lambda<N, 6, SMT::MINV_LDS>(Jh, Minv, MBinv, bie, k, tk->decoupling, Lam, Lmod, T1);

	PHASE_FENCE();
	tk = launder(tk);
	// ---- control law (MotionForceTask.cpp:286-509 with sigma_force = sigma_moment = 0, OTG off)
	double g[6];
	{
		double xp[3], Rc[9], G24[36];  // 36: the closed-loop branch of the general law reads the sensed force / moment (entries 30..35)
		UNR for (int e = 0; e < 3; e++) xp[e] = pw[e];
		UNR for (int e = 0; e < 9; e++) Rc[e] = sm.park[cbase + 3 + e][lane];
		UNR for (int e = 0; e < 24; e++) G24[e] = PF ? sm.pf[2 * N + e][lane] : tk->goal[(size_t)e * ld + b];
		double ip[3] = {0, 0, 0}, io[3] = {0, 0, 0}, Fum[6], Ff[6] = {0, 0, 0, 0, 0, 0};
		if (track) {
			UNR for (int i = 0; i < 3; i++) {
				ip[i] = tk->integ[(size_t)i * ld + b];
				io[i] = tk->integ[(size_t)(3 + i) * ld + b];
			}
		}
		if (GL) {
			// velocity saturation, force & moment spaces with open- or closed-loop force / moment control (the passivity controller around
			// the force loop stays with the general kernel: lane_eligible in the engine); reduces to the short law for the defaults
			UNR for (int e = 24; e < 30; e++) G24[e] = tk->goal[(size_t)e * ld + b];
			const bool closed = tk->cl_force || tk->cl_moment;  // batch-uniform
			double ifm[6] = {0, 0, 0, 0, 0, 0};
			UNR for (int e = 30; e < 36; e++) G24[e] = closed ? tk->goal[(size_t)e * ld + b] : 0.0;  // sensed force / moment, sensor frame
			if (closed) {
				UNR for (int i = 0; i < 6; i++) ifm[i] = tk->integ[(size_t)(6 + i) * ld + b];
			}
			law_motion_force_general<true>(tk, vw, xp, Rc, G24, ip, io, ifm, Fum, Ff);
			if (closed && live) {  // shadow rows, committed in the epilogue like the others
				UNR for (int i = 0; i < 6; i++) tk->integ_new[(size_t)(6 + i) * ld + b] = ifm[i];
			}
		} else {
			double v[3], w[3], e3[3], tmp[3];
			UNR for (int i = 0; i < 3; i++) {
				v[i] = tk->Ppos[3 * i] * vw[0] + tk->Ppos[3 * i + 1] * vw[1] + tk->Ppos[3 * i + 2] * vw[2];
				w[i] = tk->Pori[3 * i] * vw[3] + tk->Pori[3 * i + 1] * vw[4] + tk->Pori[3 * i + 2] * vw[5];
			}
			UNR for (int i = 0; i < 3; i++) e3[i] = xp[i] - G24[i];
			UNR for (int i = 0; i < 3; i++) ip[i] += (tk->sig_p[3 * i] * e3[0] + tk->sig_p[3 * i + 1] * e3[1] + tk->sig_p[3 * i + 2] * e3[2]) * tk->dt;
			UNR for (int i = 0; i < 3; i++) tmp[i] = G24[18 + i] - tk->kp_pos[i] * e3[i] - tk->kv_pos[i] * (v[i] - G24[12 + i]) - tk->ki_pos[i] * ip[i];
			UNR for (int i = 0; i < 3; i++) Fum[i] = tk->sig_p[3 * i] * tmp[0] + tk->sig_p[3 * i + 1] * tmp[1] + tk->sig_p[3 * i + 2] * tmp[2];
			double oe[3] = {0, 0, 0};  // orientationError(desired, current) = -1/2 sum_c Rc[:,c] x Rd[:,c]
			UNR for (int c = 0; c < 3; c++) {
				const double a0 = Rc[c], a1 = Rc[3 + c], a2 = Rc[6 + c], b0 = G24[3 + c], b1 = G24[6 + c], b2 = G24[9 + c];
				oe[0] -= 0.5 * (a1 * b2 - a2 * b1);
				oe[1] -= 0.5 * (a2 * b0 - a0 * b2);
				oe[2] -= 0.5 * (a0 * b1 - a1 * b0);
			}
			double dphi[3];
			UNR for (int i = 0; i < 3; i++) dphi[i] = tk->sig_o[3 * i] * oe[0] + tk->sig_o[3 * i + 1] * oe[1] + tk->sig_o[3 * i + 2] * oe[2];
			UNR for (int i = 0; i < 3; i++) io[i] += dphi[i] * tk->dt;
			UNR for (int i = 0; i < 3; i++) tmp[i] = G24[21 + i] - tk->kp_ori[i] * dphi[i] - tk->kv_ori[i] * (w[i] - G24[15 + i]) - tk->ki_ori[i] * io[i];
			UNR for (int i = 0; i < 3; i++) Fum[3 + i] = tk->sig_o[3 * i] * tmp[0] + tk->sig_o[3 * i + 1] * tmp[1] + tk->sig_o[3 * i + 2] * tmp[2];
		}
		if (track && live) {  // shadow copy: committed in the epilogue once the instance's final status is known (a later task may still flag it)
			UNR for (int i = 0; i < 3; i++) {
				tk->integ_new[(size_t)i * ld + b] = ip[i];
				tk->integ_new[(size_t)(3 + i) * ld + b] = io[i];
			}
		}
		double fh[6], ff[6];  // fhat = Bm^T F_um, Bm^T F_f  (columns >= k of Bm are zero)
		if (tk->bm_identity) {
			UNR for (int a = 0; a < 6; a++) {
				fh[a] = Fum[a];
				ff[a] = Ff[a];
			}
		} else {
			UNR for (int a = 0; a < 6; a++) {
				double s = 0.0, s2 = 0.0;
				UNR for (int e = 0; e < 6; e++) {
					s = fma(tk->Bm[e * 6 + a], Fum[e], s);
					s2 = fma(tk->Bm[e * 6 + a], Ff[e], s2);
				}
				fh[a] = s;
				ff[a] = s2;
			}
		}
		UNR for (int a = 0; a < 6; a++) {  // g = Lambda_mod fhat + ffhat  (SingularityHandler.cpp:307-309)
			double s = ff[a];
			UNR for (int c = 0; c < 6; c++) s = fma(Lmod[a][c], fh[c], s);
			g[a] = s;
		}
		// computeTorques(tau_prec): the reference's _Lambda is never written => no compensation
	}
	PHASE_FENCE();
	apply_task<N, 6>(Jh, g, Lam, T1, k, first, last, tau, Np);
	PHASE_FENCE();
}

template <int N, int RJ, bool PF, bool GL, typename SMT>
__device__ __forceinline__ void lane_joint_task(const CycleParams& P, SMT& sm, const ModelCPtr md, const TaskCPtr tasks, const int t, const int lane, const int b,
		const bool live, const bool first, const bool last, const SymMat<N, SMT::MINV_LDS>& Minv, const SymMat<N, SMT::MINV_LDS>& MBinv,
		const BieRankOne<N>& bie, double (&Np)[N][N], double (&tau)[N], bool& singular) {
	constexpr int LD = SMT::LD;
	constexpr int P_Z = 0, P_O = 3 * N, P_CAP = 6 * N;
	const size_t ld = P.ld;
	TaskCPtr tk = launder(tasks + t);
	const bool track = (tk->has_ki || P.integ_always);
	(void)LD; (void)P_Z; (void)P_O; (void)P_CAP; (void)md;
	// ======================================================== JointTask (RJ rows in the range basis)
	const int m = tk->m;
	// control-law inputs (goal rows, joint velocities) are requested NOW so that their HBM / L2 latency is covered by the range-basis
	// and Lambda algebra below; they are consumed ~7000 clocks later (the loads cannot cross the phase fences)
	double pre_goal[3 * N], pre_dq[N];
	if (!PF) {
		UNR for (int i = 0; i < N; i++) {
			const int ii = (i < m) ? i : 0;
			pre_goal[i] = tk->goal[(size_t)ii * ld + b];
			pre_goal[N + i] = tk->goal[(size_t)(m + ii) * ld + b];
			pre_goal[2 * N + i] = tk->goal[(size_t)(2 * m + ii) * ld + b];
			pre_dq[i] = SMT::DQ_LDS ? 0.0 : P.dq[(size_t)i * ld + b];
		}
	}
	double U[N][RJ], Jh[RJ][N];
	int r_lane = 0;
	bool need_exact = false;
	{
		// Jp = S N_prec; column-pivoted Gram-Schmidt on its columns with a clean-gap acceptance rule.
		// The deflation coefficients u_s^T W are exactly the rows of Jh = U^T Jp.
		double Wc[N][N];
		if (tk->s_identity) {
			UNR for (int i = 0; i < N; i++)
				UNR for (int j = 0; j < N; j++) Wc[i][j] = Np[i][j];
		} else {
			UNR for (int i = 0; i < N; i++)
				UNR for (int j = 0; j < N; j++) {
					double s = 0.0;
					UNR for (int l = 0; l < N; l++) s = fma(tk->S[i * N + l], Np[l][j], s);  // rows >= m of the stored S are zero (host zero-fills): no branch per element
					Wc[i][j] = s;
				}
		}
		double c0 = 1.0;
		bool going = true;
		bool pending = false;  // an ambiguous pivot was met: keep sweeping, the full-row-rank certificate below decides
		UNR for (int s = 0; s <= RJ; s++) {
			double cn[N], best = -1.0, frob2 = 0.0;
			UNR for (int j = 0; j < N; j++) {
				double a = 0.0;
				UNR for (int i = 0; i < N; i++) a = fma(Wc[i][j], Wc[i][j], a);
				cn[j] = a;
				frob2 += a;
				best = fmax(best, a);
			}
			if (s == 0) {
				c0 = best;
				// ||Jp||_F < 1e-3 or sigma_0 < 1e-3 => empty range (matrixRangeBasis); sigma_0^2 lies in [best, frob2]
				if (frob2 < 1e-6) going = false;
				else if (best < 1e-5) { going = false; need_exact = true; }
			} else if (going) {
				const double ratio = best / c0;
				if (ratio < 1e-20) {  // numerically exact rank; ambiguous pivots before it are only certifiable when every row was kept
					going = false;
					need_exact = need_exact || (pending && s < m);
				} else if (s == RJ) {  // rank bound exceeded: exact path decides
					going = false;
					need_exact = true;
				} else if (ratio < 1e-4) {
					pending = true;  // ambiguous gap: continue, see the certificate after the sweep
				}
			}
			if (s < RJ) {
				double pv[N], pbest = -1.0;
				UNR for (int i = 0; i < N; i++) pv[i] = 0.0;
				UNR for (int j = 0; j < N; j++) {
					const bool take = cn[j] > pbest;
					pbest = take ? cn[j] : pbest;
					UNR for (int i = 0; i < N; i++) pv[i] = take ? Wc[i][j] : pv[i];
				}
				const double inv = going ? rsqrt(pbest) : 0.0;
				UNR for (int i = 0; i < N; i++) {
					pv[i] *= inv;
					U[i][s < RJ ? s : 0] = pv[i];
				}
				UNR for (int j = 0; j < N; j++) {
					double d = 0.0;
					UNR for (int i = 0; i < N; i++) d = fma(pv[i], Wc[i][j], d);
					Jh[s < RJ ? s : 0][j] = d;
					UNR for (int i = 0; i < N; i++) Wc[i][j] = fma(-d, pv[i], Wc[i][j]);
				}
				if (going) r_lane = s + 1;
			}
		}
		// Ambiguous pivots but every row direction kept (r == m): the reference keeps them too (matrixRangeBasis returns the
		// identity) iff sigma_min/sigma_0 >= 1e-3.  Certified by lambda_min(Jp Jp^T) > 1e-6 trace(Jp Jp^T) >= 1e-6 sigma_0^2
		// (LDL^T pivots of the shifted Gram matrix); any orthonormal basis of the full row space is then exact (invariance).
		if (__any(pending && !need_exact)) {
			bool cert = (r_lane == m);
			double G[N][N];
			{
				double Jp[N][N];
				UNR for (int i = 0; i < N; i++)
					UNR for (int j = 0; j < N; j++) {
						double sacc = 0.0;
						UNR for (int l = 0; l < N; l++) sacc = fma(tk->S[i * N + l], Np[l][j], sacc);
						Jp[i][j] = sacc;
					}
				double trg = 0.0;
				UNR for (int a = 0; a < N; a++)
					UNR for (int c = 0; c <= a; c++) {
						double sacc = 0.0;
						UNR for (int j = 0; j < N; j++) sacc = fma(Jp[a][j], Jp[c][j], sacc);
						G[a][c] = sacc;
						if (a == c) trg += sacc;
					}
				const double shift = 1e-6 * trg;
				UNR for (int a = 0; a < N; a++) G[a][a] = (a < m) ? G[a][a] - shift : 1.0;
			}
			UNR for (int kk = 0; kk < N; kk++) {
				const double dpv = G[kk][kk];
				cert = cert && (dpv > 0.0);
				const double id = 1.0 / dpv;
				UNR for (int i = kk + 1; i < N; i++) {
					const double lik = G[i][kk] * id;
					UNR for (int j = kk + 1; j <= i; j++) G[i][j] = fma(-lik, G[j][kk], G[i][j]);
				}
			}
			if (pending && !cert) need_exact = true;
		}
	}
	// exact path (rare): Gram eigen-decomposition by the whole wavefront, reference tolerance semantics
	unsigned long long need = __ballot(need_exact);
	while (need) {
		const int src = __builtin_ctzll(need);
		need &= need - 1;
		if (lane == src) {
			UNR for (int i = 0; i < N; i++)
				UNR for (int j = 0; j < N; j++) {
					double s = 0.0;
					if (i < m) {
						UNR for (int l = 0; l < N; l++) s = fma(tk->S[i * N + l], Np[l][j], s);
					}
					sm.Jp[i * LD + j] = s;
				}
		}
		__syncthreads();
		gemm<64>(sm.A, LD, sm.Jp, LD, 1, sm.Jp, 1, LD, m, N, m, lane);
		__syncthreads();
		double trc = 0.0;
		for (int i = 0; i < m; i++) trc += sm.A[i * LD + i];
		__syncthreads();
		jacobi_eig<64>(sm.A, sm.V, m, LD, sm.cs, &sm.flag, lane);
		if (lane == 0) {
			double lmax = 0.0;
			for (int i = 0; i < m; i++) lmax = fmax(lmax, sm.A[i * LD + i]);
			int rr = 0;
			if (sqrt(fmax(trc, 0.0)) >= 1e-3 && sqrt(lmax) >= 1e-3) {
				for (int i = 0; i < m; i++)
					if (sqrt(fmax(sm.A[i * LD + i], 0.0) / lmax) >= 1e-3) sm.colidx[rr++] = i;
			}
			sm.rank = rr;
		}
		__syncthreads();
		if (lane == src) {
			const int rr = sm.rank;
			if (rr > RJ) singular = true;  // cannot happen while the host's rank bound holds; never silently truncate
			r_lane = rr < RJ ? rr : RJ;
			UNR for (int c = 0; c < RJ; c++) {
				const int col = (c < rr) ? sm.colidx[c] : 0;
				UNR for (int i = 0; i < N; i++) {
					double uv = 0.0;
					if (c < rr && i < m) uv = (rr == m) ? ((i == c) ? 1.0 : 0.0) : sm.V[i * LD + col];
					U[i][c] = uv;
				}
			}
			UNR for (int c = 0; c < RJ; c++)
				UNR for (int j = 0; j < N; j++) {
					double s = 0.0;
					UNR for (int i = 0; i < N; i++) s = fma(U[i][c], sm.Jp[i * LD + j], s);
					Jh[c][j] = s;
				}
		}
		__syncthreads();
	}
	PHASE_FENCE();
	double Lam[RJ][RJ], Lmod[RJ][RJ], T1[RJ][N];
	opspace_lambda<N, RJ, SMT::MINV_LDS>(Jh, Minv, MBinv, bie, r_lane, tk->decoupling, Lam, Lmod, T1);
	DIAG_STAMP();

	tk = launder(tk);
	// ---- control law (JointTask.cpp:285-356, OTG off, no velocity saturation)
	double g[RJ];
	{
		double w[N], qv[N], dqv[N];
		UNR for (int i = 0; i < N; i++) {  // w = M^-1 tau_prec (disturbance compensation, :287-291)
			double mr[N], s = 0.0;
			Minv.row(i, mr);
			UNR for (int j = 0; j < N; j++) s = fma(mr[j], tau[j], s);
			w[i] = s;
		}
		DIAG_STAMP();
		UNR for (int j = 0; j < N; j++) {
			qv[j] = PF ? sm.pf[j][lane] : sm.qsh[j][lane];  // the joint angles are still parked in LDS from the kinematics
			dqv[j] = PF ? sm.pf[N + j][lane] : (SMT::DQ_LDS ? sm.dqsh[j][lane] : pre_dq[j]);
		}
		double a1[RJ], b1[RJ];
		UNR for (int c = 0; c < RJ; c++) a1[c] = b1[c] = 0.0;
		// straight-line over all N rows: rows >= m of U are zero (zero rows of S), so they drop out of a1 / b1 without a branch per row;
		// only the integrator traffic is guarded
		const bool s_id = tk->s_identity != 0;
		const double dt_ = tk->dt;
		UNR for (int i = 0; i < N; i++) {
			double cur = 0.0, vel = 0.0, sw = 0.0;
			if (s_id) {
				cur = qv[i];
				vel = dqv[i];
				sw = w[i];
			} else {
				UNR for (int j = 0; j < N; j++) {
					const double sij = tk->S[i * N + j];
					cur = fma(sij, qv[j], cur);
					vel = fma(sij, dqv[j], vel);
					sw = fma(sij, w[j], sw);
				}
			}
			const double e = cur - (PF ? sm.pf[2 * N + 24 + i][lane] : pre_goal[i]);
			double ie = 0.0;
			if (track && i < m) ie = tk->integ[(size_t)i * ld + b];
			ie += e * dt_;  // :323-324
			if (track && live && i < m) tk->integ_new[(size_t)i * ld + b] = ie;
			double fi;
			if (GL && tk->vel_sat) {  // JointTask.cpp:327-341
				double vdes = -tk->kp[i] * tk->kvinv[i] * e - tk->ki[i] * tk->kvinv[i] * ie;
				vdes = fmin(fmax(vdes, -tk->sat[i]), tk->sat[i]);
				fi = -tk->kv[i] * (vel - vdes);
			} else {
				fi = -tk->kp[i] * e - tk->kv[i] * (vel - (PF ? sm.pf[2 * N + 24 + N + i][lane] : pre_goal[N + i])) - tk->ki[i] * ie;  // :342-345
			}
			const double ai = (PF ? sm.pf[2 * N + 24 + 2 * N + i][lane] : pre_goal[2 * N + i]) - sw;
			UNR for (int c = 0; c < RJ; c++) {
				a1[c] = fma(U[i][c], ai, a1[c]);
				b1[c] = fma(U[i][c], fi, b1[c]);
			}
		}
		UNR for (int a = 0; a < RJ; a++) {  // g = Lambda a1 + Lambda_mod b1  (:348-351)
			double s = 0.0;
			UNR for (int c = 0; c < RJ; c++) s = fma(Lam[a][c], a1[c], fma(Lmod[a][c], b1[c], s));
			g[a] = (a < r_lane) ? s : 0.0;
		}
	}
	DIAG_STAMP();
	apply_task<N, RJ>(Jh, g, Lam, T1, r_lane, first, last, tau, Np);
}

enum { SIG_GENERIC = 0, SIG_MF_JT = 1 };

// GL: general control laws (velocity saturation, force/moment spaces) compiled in; GL = false keeps the default-law fast path lean
template <int N, int RJ, int SIG, bool PF, bool LEAN, bool GL>
__global__ void __launch_bounds__(64) saip_cycle_lane(const CycleParams P) {
	static_assert(!PF || SIG == SIG_MF_JT, "the prefetch layout is defined for the MotionForce+Joint stack");
	using SM = LaneSmem<N, PF, LEAN>;
	constexpr int LD = SM::LD;
	constexpr int P_Z = 0, P_O = 3 * N, P_CAP = 6 * N, P_BODY = 6 * N + 24;
	__shared__ SM sm;
	const int lane = threadIdx.x;
	const int act = P.lane_active;  // instances per wavefront (64 = full)
	if (lane >= act) return;
	int b = blockIdx.x * act + lane;
	const bool live = b < P.B;
	if (!live) b = P.B - 1;  // dead lanes shadow the last instance and never store
	const ModelCPtr md = (ModelCPtr)P.model;
	const TaskCPtr tasks = (TaskCPtr)P.tasks;
	const size_t ld = P.ld;
#ifdef SAIP_PREWARM  // scalar-cache prewarm: helped the first versions, costs 1 % now (kept for experiments)
	{
		// warm the scalar cache: one double of every 64-byte line of the per-joint records and of the head of the task constants,
		// all loads in flight at once (one exposed L2 latency) instead of one cold miss per FK iteration / per task phase
		double acc = 0.0;
		UNR for (int j = 0; j < N; j++) {
			const JointRec CONSTANT_AS* jr = md->jrec + j;
			acc += jr->R0[0] + jr->R0[8] + jr->inertia[0] + jr->mass;
		}
		for (int t = 0; t < P.nt; t++) {
			const TaskCPtr tk = tasks + t;
			acc += tk->dt + tk->rot[0] + tk->Bm[0] + tk->Bm[8] + tk->Bm[16] + tk->Bm[24] + tk->Bm[32] + tk->Ppos[4] + tk->sig_p[0] + tk->sig_o[0] +
				   tk->kp_pos[0] + tk->kp_ori[2] + tk->kp[0] + tk->kp[8] + tk->kv[0] + tk->ki[0];
		}
		asm volatile("" ::"s"(acc));
	}
#endif
	PHASE_FENCE();

	// ---------------------------------------------------------------- latency variant: every per-instance input is fetched up front
	// (59 independent coalesced loads in flight at once, one wait) and parked in LDS; nothing below touches HBM again
	// until the torque store.  Used when the batch leaves at most one wavefront per CU (LDS is free then).
	if (PF) {
		double v[2 * N + 24 + 3 * N];
		UNR for (int j = 0; j < N; j++) {
			v[j] = P.q[(size_t)j * ld + b];
			v[N + j] = P.dq[(size_t)j * ld + b];
		}
		UNR for (int e = 0; e < 24; e++) v[2 * N + e] = tasks[0].goal[(size_t)e * ld + b];
		UNR for (int e = 0; e < 3 * N; e++) v[2 * N + 24 + e] = tasks[1].goal[(size_t)e * ld + b];  // full joint task: m == N
		UNR for (int e = 0; e < 2 * N + 24 + 3 * N; e++) sm.pf[e][lane] = v[e];
	}
	PHASE_FENCE();
	// ---------------------------------------------------------------- forward kinematics (ONE rolled loop over the joints: one code copy,
	// short live ranges); joint axes/origins, per-body spatial inertias and the control frames are parked in LDS
	double Minv[N][N], MBinv[N][N], taug[N];
	BieRankOne<N> bie;
	bie.coef = 0.0;
	bie.use_sm = false;
	UNR for (int i = 0; i < N; i++) bie.v[i] = 0.0;
	{
		double Mm[N][N];
		if (LEAN) {
			UNR for (int i = 0; i < N; i++) {
				taug[i] = 0.0;
				UNR for (int c = 0; c < N; c++) Mm[i][c] = 0.0;
			}
		}
		{
			double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
			if (!PF) {  // all joint angles in flight at once (one exposed HBM latency instead of one per loop iteration)
				double qv[N], dv[SM::DQ_LDS ? N : 1];
				UNR for (int j = 0; j < N; j++) qv[j] = P.q[(size_t)j * ld + b];
				if (SM::DQ_LDS) {
					UNR for (int j = 0; j < N; j++) dv[j] = P.dq[(size_t)j * ld + b];
				}
				UNR for (int j = 0; j < N; j++) sm.qsh[j][lane] = qv[j];
				if (SM::DQ_LDS) {
					UNR for (int j = 0; j < N; j++) sm.dqsh[j][lane] = dv[j];
				}
			}
			_Pragma("unroll 1") for (int j = 0; j < N; j++) {
				const JointRec CONSTANT_AS* jr = md->jrec + j;
				const double CONSTANT_AS* R0 = jr->R0;
				const double CONSTANT_AS* p0 = jr->p0;
				const double ax = jr->axis[0], ay = jr->axis[1], az = jr->axis[2];
				const double qj = PF ? sm.pf[j][lane] : sm.qsh[j][lane];
				double Rt[9];
				UNR for (int r = 0; r < 3; r++) o[r] += R[3 * r] * p0[0] + R[3 * r + 1] * p0[1] + R[3 * r + 2] * p0[2];
				UNR for (int r = 0; r < 3; r++)
					UNR for (int c = 0; c < 3; c++) Rt[3 * r + c] = R[3 * r] * R0[c] + R[3 * r + 1] * R0[3 + c] + R[3 * r + 2] * R0[6 + c];
				if (jr->jtype == 1) {
					double s, c;
					sincos_joint(qj, &s, &c);
					if (jr->axis_is_z) {  // rotation about the local z axis mixes the first two columns only
						UNR for (int r = 0; r < 3; r++) {
							R[3 * r] = fma(c, Rt[3 * r], s * Rt[3 * r + 1]);
							R[3 * r + 1] = fma(c, Rt[3 * r + 1], -s * Rt[3 * r]);
							R[3 * r + 2] = Rt[3 * r + 2];
						}
					} else {
						const double v = 1.0 - c;
						const double Rq[9] = {c + ax * ax * v,      ax * ay * v - az * s, ax * az * v + ay * s,
											  ay * ax * v + az * s, c + ay * ay * v,      ay * az * v - ax * s,
											  az * ax * v - ay * s, az * ay * v + ax * s, c + az * az * v};
						UNR for (int r = 0; r < 3; r++)
							UNR for (int cc = 0; cc < 3; cc++) R[3 * r + cc] = Rt[3 * r] * Rq[cc] + Rt[3 * r + 1] * Rq[3 + cc] + Rt[3 * r + 2] * Rq[6 + cc];
					}
				} else {
					UNR for (int r = 0; r < 3; r++) o[r] += (Rt[3 * r] * ax + Rt[3 * r + 1] * ay + Rt[3 * r + 2] * az) * qj;
					UNR for (int e = 0; e < 9; e++) R[e] = Rt[e];
				}
				UNR for (int r = 0; r < 3; r++) {
					sm.park[P_Z + 3 * j + r][lane] = R[3 * r] * ax + R[3 * r + 1] * ay + R[3 * r + 2] * az;
					sm.park[P_O + 3 * j + r][lane] = o[r];
				}
				{  // spatial inertia of body j about the world origin: m, h = m c, Io (xx yy zz xy xz yz)
					const double CONSTANT_AS* I6 = jr->inertia;
					const double Il[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
					double Iw[6], c3[3];
					if (jr->iso_inertia) {  // i * identity is invariant under the body rotation
						Iw[0] = Iw[1] = Iw[2] = I6[0];
						Iw[3] = Iw[4] = Iw[5] = 0.0;
					} else {
						double RI[9];
						UNR for (int r = 0; r < 3; r++)
							UNR for (int cc = 0; cc < 3; cc++) RI[3 * r + cc] = R[3 * r] * Il[cc] + R[3 * r + 1] * Il[3 + cc] + R[3 * r + 2] * Il[6 + cc];
						Iw[0] = RI[0] * R[0] + RI[1] * R[1] + RI[2] * R[2];
						Iw[1] = RI[3] * R[3] + RI[4] * R[4] + RI[5] * R[5];
						Iw[2] = RI[6] * R[6] + RI[7] * R[7] + RI[8] * R[8];
						Iw[3] = RI[0] * R[3] + RI[1] * R[4] + RI[2] * R[5];
						Iw[4] = RI[0] * R[6] + RI[1] * R[7] + RI[2] * R[8];
						Iw[5] = RI[3] * R[6] + RI[4] * R[7] + RI[5] * R[8];
					}
					UNR for (int r = 0; r < 3; r++) c3[r] = o[r] + R[3 * r] * jr->com[0] + R[3 * r + 1] * jr->com[1] + R[3 * r + 2] * jr->com[2];
					const double m = jr->mass, cc2 = c3[0] * c3[0] + c3[1] * c3[1] + c3[2] * c3[2];
					const double bd[10] = {m, m * c3[0], m * c3[1], m * c3[2], Iw[0] + m * (cc2 - c3[0] * c3[0]), Iw[1] + m * (cc2 - c3[1] * c3[1]),
										   Iw[2] + m * (cc2 - c3[2] * c3[2]), Iw[3] - m * c3[0] * c3[1], Iw[4] - m * c3[0] * c3[2], Iw[5] - m * c3[1] * c3[2]};
					if (!LEAN) {
						UNR for (int e = 0; e < 10; e++) sm.park[P_BODY + 10 * j + e][lane] = bd[e];
					} else {
						// throughput variant (no LDS for the bodies => 4 wavefronts per CU): body j's inertia is added to every
						// M_ac with a <= c <= j right away (O(n^3) instead of the composite sweep's O(n^2))
						double Sw[N][3], Sv[N][3];
						UNR for (int c = 0; c < N; c++) {
							if (c <= j) {
								double z[3], oo[3];
								UNR for (int e = 0; e < 3; e++) {
									z[e] = sm.park[P_Z + 3 * c + e][lane];
									oo[e] = sm.park[P_O + 3 * c + e][lane];
								}
								if (md->jtype[c] == 1) {
									UNR for (int e = 0; e < 3; e++) Sw[c][e] = z[e];
									Sv[c][0] = oo[1] * z[2] - oo[2] * z[1];
									Sv[c][1] = oo[2] * z[0] - oo[0] * z[2];
									Sv[c][2] = oo[0] * z[1] - oo[1] * z[0];
								} else {
									UNR for (int e = 0; e < 3; e++) {
										Sw[c][e] = 0.0;
										Sv[c][e] = z[e];
									}
								}
							}
						}
						const double hx = bd[1], hy = bd[2], hz = bd[3];
						UNR for (int c = 0; c < N; c++) {
							if (c <= j) {
								const double* w = Sw[c];
								const double* v = Sv[c];
								const double f[3] = {m * v[0] + (w[1] * hz - w[2] * hy), m * v[1] + (w[2] * hx - w[0] * hz), m * v[2] + (w[0] * hy - w[1] * hx)};
								const double nO[3] = {bd[4] * w[0] + bd[7] * w[1] + bd[8] * w[2] + (hy * v[2] - hz * v[1]),
													  bd[7] * w[0] + bd[5] * w[1] + bd[9] * w[2] + (hz * v[0] - hx * v[2]),
													  bd[8] * w[0] + bd[9] * w[1] + bd[6] * w[2] + (hx * v[1] - hy * v[0])};
								UNR for (int a = 0; a <= c; a++)
									Mm[a][c] += Sw[a][0] * nO[0] + Sw[a][1] * nO[1] + Sw[a][2] * nO[2] + Sv[a][0] * f[0] + Sv[a][1] * f[1] + Sv[a][2] * f[2];
							}
						}
						if (P.gravity_comp) {
							const double gx = md->gravity[0], gy = md->gravity[1], gz = md->gravity[2];
							const double hxg[3] = {hy * gz - hz * gy, hz * gx - hx * gz, hx * gy - hy * gx};
							UNR for (int a = 0; a < N; a++)
								if (a <= j) taug[a] -= Sw[a][0] * hxg[0] + Sw[a][1] * hxg[1] + Sw[a][2] * hxg[2] + m * (Sv[a][0] * gx + Sv[a][1] * gy + Sv[a][2] * gz);
						}
					}
				}
				UNR for (int s = 0; s < 2; s++) {  // control frames of the motion-force tasks attached to this body
					const int t = P.mf_task[s];
					if (t >= 0 && tasks[t].body == j) {
						const TaskCPtr tk = tasks + t;
						UNR for (int r = 0; r < 3; r++)
							sm.park[P_CAP + 12 * s + r][lane] = o[r] + R[3 * r] * tk->pos[0] + R[3 * r + 1] * tk->pos[1] + R[3 * r + 2] * tk->pos[2];
						UNR for (int r = 0; r < 3; r++)
							UNR for (int c = 0; c < 3; c++)
								sm.park[P_CAP + 12 * s + 3 + 3 * r + c][lane] = R[3 * r] * tk->rot[c] + R[3 * r + 1] * tk->rot[3 + c] + R[3 * r + 2] * tk->rot[6 + c];
					}
				}
			}
		}
		PHASE_FENCE();
		// M(q) from composite inertias (backward sweep): M_ij = S_i . (Ic_j S_j), i <= j
		if (LEAN) {
			UNR for (int c = 0; c < N; c++)
				UNR for (int a = 0; a < c; a++) Mm[c][a] = Mm[a][c];
		} else {
			double Sw[N][3], Sv[N][3];
			UNR for (int j = 0; j < N; j++) {
				double z[3], o[3];
				UNR for (int e = 0; e < 3; e++) {
					z[e] = sm.park[P_Z + 3 * j + e][lane];
					o[e] = sm.park[P_O + 3 * j + e][lane];
				}
				{  // unit motion of joint j about the world origin, branch-free (a branch per joint splits the unrolled code into tiny blocks)
					const bool rev = md->jtype[j] == 1;
					const double cx = o[1] * z[2] - o[2] * z[1], cy = o[2] * z[0] - o[0] * z[2], cz = o[0] * z[1] - o[1] * z[0];
					UNR for (int e = 0; e < 3; e++) Sw[j][e] = rev ? z[e] : 0.0;
					Sv[j][0] = rev ? cx : z[0];
					Sv[j][1] = rev ? cy : z[1];
					Sv[j][2] = rev ? cz : z[2];
				}
			}
			double Ic[10];
			UNR for (int e = 0; e < 10; e++) Ic[e] = 0.0;
			UNR for (int jj = 0; jj < N; jj++) {
				const int j = N - 1 - jj;
				PHASE_FENCE();
				UNR for (int e = 0; e < 10; e++) Ic[e] += sm.park[P_BODY + 10 * j + e][lane];
				const double* w = Sw[j];
				const double* v = Sv[j];
				const double m = Ic[0], hx = Ic[1], hy = Ic[2], hz = Ic[3];
				const double f[3] = {m * v[0] + (w[1] * hz - w[2] * hy), m * v[1] + (w[2] * hx - w[0] * hz), m * v[2] + (w[0] * hy - w[1] * hx)};
				const double nO[3] = {Ic[4] * w[0] + Ic[7] * w[1] + Ic[8] * w[2] + (hy * v[2] - hz * v[1]),
									  Ic[7] * w[0] + Ic[5] * w[1] + Ic[9] * w[2] + (hz * v[0] - hx * v[2]),
									  Ic[8] * w[0] + Ic[9] * w[1] + Ic[6] * w[2] + (hx * v[1] - hy * v[0])};
				UNR for (int i = 0; i <= j; i++) {
					const double val = Sw[i][0] * nO[0] + Sw[i][1] * nO[1] + Sw[i][2] * nO[2] + Sv[i][0] * f[0] + Sv[i][1] * f[1] + Sv[i][2] * f[2];
					Mm[i][j] = val;
					Mm[j][i] = val;
				}
				// jointGravityVector: -S_j . (gravity wrench of the subtree about the world origin) = -(w.(h x g) + v.(m g))
				if (P.gravity_comp) {  // wave-uniform; keeps the computation here instead of being sunk to the epilogue
					const double gx = md->gravity[0], gy = md->gravity[1], gz = md->gravity[2];
					const double hxg[3] = {hy * gz - hz * gy, hz * gx - hx * gz, hx * gy - hy * gx};
					taug[j] = -(w[0] * hxg[0] + w[1] * hxg[1] + w[2] * hxg[2] + m * (v[0] * gx + v[1] * gy + v[2] * gz));
				} else {
					taug[j] = 0.0;
				}
			}
		}
		PHASE_FENCE();
		spd_inv_reg<N>(Mm, Minv);
		PHASE_FENCE();
		if (P.any_bie) {  // M_BIE: diagonal clamped from below (SingularityHandler.cpp:176-182, JointTask.cpp:254-259)
			double dsum = 0.0, dd = 0.0;
			int nclamp = 0;
			UNR for (int i = 0; i < N; i++) bie.v[i] = 0.0;
			UNR for (int c = 0; c < N; c++) {
				const double delta = fmax(P.bie_thr - Mm[c][c], 0.0);
				nclamp += (delta > 0.0) ? 1 : 0;
				dsum += delta;
				dd = fma(delta, Minv[c][c], dd);
				UNR for (int i = 0; i < N; i++) bie.v[i] = fma(delta, Minv[i][c], bie.v[i]);
			}
			bie.coef = (dsum > 0.0) ? 1.0 / (dsum * (1.0 + dd)) : 0.0;
			bie.use_sm = !__any(nclamp > 1);
			if (!bie.use_sm) {
				UNR for (int i = 0; i < N; i++) Mm[i][i] = fmax(Mm[i][i], P.bie_thr);
				spd_inv_reg<N>(Mm, MBinv);
			} else {
				UNR for (int i = 0; i < N; i++)
					UNR for (int j = 0; j < N; j++) MBinv[i][j] = Minv[i][j];  // not read on the rank-one path
			}
		} else {
			UNR for (int i = 0; i < N; i++)
				UNR for (int j = 0; j < N; j++) MBinv[i][j] = Minv[i][j];
		}
	}

	constexpr int NSYM = N * (N + 1) / 2;
	if (SM::MINV_LDS) {
		UNR for (int i = 0; i < N; i++)
			UNR for (int j = 0; j <= i; j++) {
				sm.minv[i * (i + 1) / 2 + j][lane] = Minv[i][j];
				sm.minv[NSYM + i * (i + 1) / 2 + j][lane] = MBinv[i][j];
			}
	}
	const SymMat<N, SM::MINV_LDS> MinvS = {Minv, SM::MINV_LDS ? &sm.minv[0] : nullptr, lane};
	const SymMat<N, SM::MINV_LDS> MBinvS = {MBinv, SM::MINV_LDS ? &sm.minv[SM::MINV_LDS ? NSYM : 0] : nullptr, lane};
	PHASE_FENCE();
	// ---------------------------------------------------------------- task hierarchy
	double Np[N][N], tau[N];
	UNR for (int i = 0; i < N; i++) {
		tau[i] = 0.0;
		UNR for (int j = 0; j < N; j++) Np[i][j] = (i == j) ? 1.0 : 0.0;
	}
	bool singular = false;
	bool first = true;

	if (SIG == SIG_MF_JT) {
		// the common stack (motion-force task + joint task in its nullspace), straight-line: no loop-carried N_prec
		lane_motion_force_task<N, PF, GL, SM>(P, sm, md, tasks, 0, lane, b, live, true, false, MinvS, MBinvS, bie, Np, tau, singular);
		PHASE_FENCE();
		lane_joint_task<N, RJ, PF, GL, SM>(P, sm, md, tasks, 1, lane, b, live, false, true, MinvS, MBinvS, bie, Np, tau, singular);
	} else {
		_Pragma("unroll 1") for (int t = 0; t < P.nt; t++) {
			const bool last = (t == P.nt - 1);
			if (tasks[t].type == TASK_MOTION_FORCE) lane_motion_force_task<N, PF, GL, SM>(P, sm, md, tasks, t, lane, b, live, first, last, MinvS, MBinvS, bie, Np, tau, singular);
			else lane_joint_task<N, RJ, PF, GL, SM>(P, sm, md, tasks, t, lane, b, live, first, last, MinvS, MBinvS, bie, Np, tau, singular);
			first = false;
		}
	}

	// ---------------------------------------------------------------- RobotController post-processing (RobotController.cpp:86-116) + store
	if (P.torque_sat) {
		UNR for (int j = 0; j < N; j++) {
			const double lim = md->effort[j];
			tau[j] = tau[j] > lim ? lim : (tau[j] < -lim ? -lim : tau[j]);
		}
	}
	if (P.jla) {
		// tau = JLA.computeTorques(tau) + N_constraints^T tau with N_constraints^T = I - S^T (S M^-1 S^T)^-1 S M^-1, S = rows of the
		// identity for the joints inside a limit zone (JointLimitAvoidanceTask.cpp:124-169 with N_prec = I).  Padded to N x N.
		double lat[N];
		bool act[N], any = false;
		UNR for (int j = 0; j < N; j++) {
			const double qj = P.q[(size_t)j * ld + b], dqj = P.dq[(size_t)j * ld + b];
			lat[j] = jla_joint(qj, dqj, md->q_lower[j], md->q_upper[j], md->vel_limit[j], md->effort[j], tau[j], &act[j]);
			any = any || act[j];
		}
		if (__any(any)) {  // wave-uniform: free when no instance of the wavefront touches a limit zone
			double w[N], A[N][N], Ai[N][N];
			UNR for (int i = 0; i < N; i++) {
				double mr[N], sacc = 0.0;
				MinvS.row(i, mr);
				UNR for (int j = 0; j < N; j++) sacc = fma(mr[j], tau[j], sacc);
				w[i] = act[i] ? sacc : 0.0;
				UNR for (int j = 0; j <= i; j++) A[i][j] = (act[i] && act[j]) ? mr[j] : ((i == j) ? 1.0 : 0.0);
			}
			spd_inv_reg<N>(A, Ai);
			UNR for (int i = 0; i < N; i++) {
				double y = 0.0;
				UNR for (int j = 0; j < N; j++) y = fma(Ai[i][j], w[j], y);
				tau[i] = act[i] ? (lat[i] + tau[i] - y) : tau[i];
			}
			if (P.torque_sat) {
				UNR for (int j = 0; j < N; j++) {
					const double lim = md->effort[j];
					tau[j] = tau[j] > lim ? lim : (tau[j] < -lim ? -lim : tau[j]);
				}
			}
		}
	}
	PHASE_FENCE();
	if (live) {
		UNR for (int j = 0; j < N; j++) {
			double tv = tau[j];
			if (P.gravity_comp) tv += taug[j];
			// a flagged instance holds its last valid torques (or gets them from the slow path launched behind) unless NaN was asked for
			if (!singular) P.tau[(size_t)j * ld + b] = tv;
			else if (P.flag_nan) P.tau[(size_t)j * ld + b] = __builtin_nan("");
		}
		P.status[b] = singular ? 1 : 0;
		if (singular && P.flag_list) {
			const int slot = atomicAdd(P.flag_count, 1);
			if (slot < P.B) P.flag_list[slot] = b;
		}  // work list of the device-side slow path
		if (P.flag_list && blockIdx.x == 0 && lane == 0) *P.flag_count_next = 0;   // the other list: its last readers finished a cycle ago
		// integrators: commit what the control laws advanced them to, unless the instance ended flagged (then nothing advances: the slow
		// path launched behind recomputes it from the untouched state, or it stays invalid)
		if (!singular) {
			_Pragma("unroll 1") for (int t = 0; t < P.nt; t++) {
				const TaskCPtr tk = tasks + t;
				if (tk->type == TASK_MOTION_FORCE && (tk->cl_force || tk->cl_moment)) {  // force / moment integrators (rows 6..11)
					_Pragma("unroll 1") for (int i = 6; i < 12; i++) tk->integ[(size_t)i * ld + b] = tk->integ_new[(size_t)i * ld + b];
				}
				if (!(tk->has_ki || P.integ_always)) continue;
				const int rows = tk->type == TASK_MOTION_FORCE ? 6 : tk->m;
				_Pragma("unroll 1") for (int i = 0; i < rows; i++) tk->integ[(size_t)i * ld + b] = tk->integ_new[(size_t)i * ld + b];
			}
		}
	}
	PHASE_FENCE();
}

// Smallest instantiated joint-rank bound >= rj for each dof; 0 = not covered.
static int pick_rj(int n, int rj) {
	if (n == 7) return rj <= 1 ? 1 : (rj <= 4 ? 4 : 7);
	if (n == 8) return rj <= 2 ? 2 : 8;
	if (n == 6) return 6;
	return 0;
}

#if defined(SAIP_LANE_LEAN_TU)
// Translation unit of the throughput (LEAN) instantiation only: built with the default machine scheduler, under which it is 6 %
// faster than under max-ilp (the latency variants are the other way round); see capi.build_library.
hipError_t launch_cycle_lane_lean(const CycleParams& P, hipStream_t stream) {
	const dim3 grid((P.B + P.lane_active - 1) / P.lane_active), block(64);
	hipLaunchKernelGGL((saip_cycle_lane<7, 1, SIG_MF_JT, false, true, false>), grid, block, 0, stream, P);
	return hipGetLastError();
}
#else
hipError_t launch_cycle_lane_lean(const CycleParams& P, hipStream_t stream);

hipError_t launch_cycle_lane(const CycleParams& P, hipStream_t stream, bool* supported) {
	int rj = 1;
	for (int t = 0; t < P.nt; t++)
		if (P.lane_task_is_joint[t] && P.rank_bound[t] > rj) rj = P.rank_bound[t];
	const int inst = pick_rj(P.n, rj);
	*supported = inst != 0;
	if (!inst) return hipSuccess;
	const bool mf_jt = (P.nt == 2 && !P.lane_task_is_joint[0] && P.lane_task_is_joint[1]);
	// (a variant that prefetched every input into LDS at kernel entry, template parameter PF, measured no faster -- 22.9 vs 22.8 us at
	// cfg2 -- and is no longer instantiated; the joint-task inputs are requested early in registers instead, see lane_joint_task)
	// throughput variant once the batch needs more than one wavefront per CU: 36 KB LDS per wavefront instead of 100 KB
	static const char* lean_env = getenv("SAIP_LANE_LEAN");  // "0"/"1" forces the choice (profiling)
	const bool lean = lean_env ? (lean_env[0] == '1') : (P.B > 64 * 256);
	const dim3 grid((P.B + P.lane_active - 1) / P.lane_active), block(64);
#define LAUNCH(NN, RR, SS, PP, GG) hipLaunchKernelGGL((saip_cycle_lane<NN, RR, SS, PP, false, GG>), grid, block, 0, stream, P)
	const bool gl = P.lane_general != 0;
	if (P.n == 7 && inst == 1 && mf_jt && !gl && lean) return launch_cycle_lane_lean(P, stream);
	else if (P.n == 7 && inst == 1 && mf_jt && !gl) LAUNCH(7, 1, SIG_MF_JT, false, false);
#ifndef SAIP_LANE_ONLY_71
	else if (P.n == 7 && inst == 1 && mf_jt) LAUNCH(7, 1, SIG_MF_JT, false, true);
	else if (P.n == 7 && inst == 1) LAUNCH(7, 1, SIG_GENERIC, false, true);
	else if (P.n == 7 && inst == 4 && mf_jt && !gl) LAUNCH(7, 4, SIG_MF_JT, false, false);
	else if (P.n == 7 && inst == 4 && mf_jt) LAUNCH(7, 4, SIG_MF_JT, false, true);
	else if (P.n == 7 && inst == 4) LAUNCH(7, 4, SIG_GENERIC, false, true);
	else if (P.n == 7) LAUNCH(7, 7, SIG_GENERIC, false, true);
	else if (P.n == 8 && inst == 2) LAUNCH(8, 2, SIG_GENERIC, false, true);
	else if (P.n == 8) LAUNCH(8, 8, SIG_GENERIC, false, true);
	else LAUNCH(6, 6, SIG_GENERIC, false, true);
#else
	else { *supported = false; return hipSuccess; }
#endif
#undef LAUNCH
	return hipGetLastError();
}
#endif  // SAIP_LANE_LEAN_TU

#if defined(SAIP_STAMP)
extern "C" int saip_debug_stamps(unsigned long long* out, int n) {
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(saip_stamps), sizeof(unsigned long long) * (n < 256 ? n : 256));
}
#endif
}  // namespace saip
