// Body of the eight-lanes-per-instance forward dynamics + integration step (see saip_dynamics_oct.hip for the method); a header so that
// saip_otg.hip can put it into one launch with the next period's trajectory generation.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "saip_device.h"
#include "saip_fk.h"
#include "saip_oct_common.h"

namespace saip {

__device__ __forceinline__ void oct_wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct OctDynInst {
	double zo[8][6];   // joint motion axes about the world origin: angular part s_w (3), linear part s_v (3)
	double M[8][8];
	double rhs[8];
	double pad_[10];   // instance stride = 2 (mod 32) doubles
};
static_assert(sizeof(OctDynInst) % 256 == 16, "instance stride must be 2 (mod 32) doubles");

#define OCT_CROSS(a, b) {(a)[1] * (b)[2] - (a)[2] * (b)[1], (a)[2] * (b)[0] - (a)[0] * (b)[2], (a)[0] * (b)[1] - (a)[1] * (b)[0]}

// inclusive prefix (towards higher joints) / suffix (towards lower joints) sums of K doubles over the 8 lanes of an instance
// (interleaved lane layout of saip_oct_common.h: the scans need no masking)
template <int K>
__device__ __forceinline__ void oct_prefix_sum(double (&x)[K], const int r) {
#define OCT_PSUM_STEP(D)                                                                              \
	UNR for (int e = 0; e < K; e++) {                                                                 \
		const double v = OCT_ILV ? octl_shr<D, false>(x[e]) : oct_dpp<DPP_ROW_SHR + D>(x[e]);         \
		x[e] += (OCT_ILV || r >= D) ? v : 0.0;                                                        \
	}
	OCT_PSUM_STEP(1)
	OCT_PSUM_STEP(2)
	OCT_PSUM_STEP(4)
#undef OCT_PSUM_STEP
}
template <int K>
__device__ __forceinline__ void oct_suffix_sum(double (&x)[K], const int r) {
#define OCT_SSUM_STEP(D)                                                                              \
	UNR for (int e = 0; e < K; e++) {                                                                 \
		const double v = OCT_ILV ? octl_shl<D>(x[e]) : oct_dpp<DPP_ROW_SHL + D>(x[e]);                \
		x[e] += (OCT_ILV || r + D < 8) ? v : 0.0;                                                     \
	}
	OCT_SSUM_STEP(1)
	OCT_SSUM_STEP(2)
	OCT_SSUM_STEP(4)
#undef OCT_SSUM_STEP
}

// one group of eight instances (block8 = its index) by the calling wavefront; smem = that wavefront's eight LDS blocks.  All
// synchronisation is wavefront-local (LDS operations of one wavefront execute in order), so the body can sit in workgroups of any
// number of wavefronts -- saip_integrate_oct (one) and the fused integrate + next-period OTG launch of saip_otg.hip (four).
__device__ __forceinline__ void integrate_oct_body(const SimParams& S, OctDynInst* smem, const int block8, const int lane) {
	constexpr int N = 7;
	const int r = octl_r(lane), grp = octl_grp(lane);
	const int rr = r < N ? r : N - 1;
	const bool act = r < N;
	OctDynInst& sm = smem[grp];
	const int b_raw = block8 * 8 + grp;
	const bool live = b_raw < S.B;
	const int b = live ? b_raw : S.B - 1;
	const size_t ld = S.ld;
	const ModelDev& md = *S.model;

	double q_r = S.q[(size_t)rr * ld + b];
	double dq_r = S.dq[(size_t)rr * ld + b];
	double tau_r = S.tau[(size_t)rr * ld + b];
	if (!(tau_r == tau_r)) tau_r = 0.0;  // flagged instances (NaN torques) coast
	double ddq_r = 0.0;
	const bool rev = md.jtype[rr] == 1;
	const double R0[9] = {md.R0[rr][0], md.R0[rr][1], md.R0[rr][2], md.R0[rr][3], md.R0[rr][4], md.R0[rr][5], md.R0[rr][6], md.R0[rr][7], md.R0[rr][8]};
	const double p0[3] = {md.p0[rr][0], md.p0[rr][1], md.p0[rr][2]};
	const double ax[3] = {md.axis[rr][0], md.axis[rr][1], md.axis[rr][2]};
	const double cl[3] = {md.com[rr][0], md.com[rr][1], md.com[rr][2]};
	const double I6[6] = {md.inertia[rr][0], md.inertia[rr][1], md.inertia[rr][2], md.inertia[rr][3], md.inertia[rr][4], md.inertia[rr][5]};
	const double mass = act ? md.mass[rr] : 0.0;

	for (int step = 0; step < S.substeps; step++) {
		// ---- kinematics: local transform, prefix product over the chain (DPP)
		double Tw[12];
		if (rev) {
			double s, c;
			sincos_joint(q_r, &s, &c);
			const double v = 1.0 - c;
			const double Rq[9] = {c + ax[0] * ax[0] * v,         ax[0] * ax[1] * v - ax[2] * s, ax[0] * ax[2] * v + ax[1] * s,
								  ax[1] * ax[0] * v + ax[2] * s, c + ax[1] * ax[1] * v,         ax[1] * ax[2] * v - ax[0] * s,
								  ax[2] * ax[0] * v - ax[1] * s, ax[2] * ax[1] * v + ax[0] * s, c + ax[2] * ax[2] * v};
			UNR for (int i = 0; i < 3; i++)
				UNR for (int j = 0; j < 3; j++) Tw[3 * i + j] = R0[3 * i] * Rq[j] + R0[3 * i + 1] * Rq[3 + j] + R0[3 * i + 2] * Rq[6 + j];
			UNR for (int e = 0; e < 3; e++) Tw[9 + e] = p0[e];
		} else {
			UNR for (int e = 0; e < 9; e++) Tw[e] = R0[e];
			double d[3];
			oct_mat3_vec(R0, ax, d);
			UNR for (int e = 0; e < 3; e++) Tw[9 + e] = p0[e] + d[e] * q_r;
		}
	OCT_FRAME_STEP(1, Tw, r)
	OCT_FRAME_STEP(2, Tw, r)
	OCT_FRAME_STEP(4, Tw, r)
		double z[3], o[3] = {Tw[9], Tw[10], Tw[11]};
		oct_mat3_vec(Tw, ax, z);
		// joint motion axis about the world origin
		const double oxz[3] = OCT_CROSS(o, z);
		const double sw[3] = {rev ? z[0] : 0.0, rev ? z[1] : 0.0, rev ? z[2] : 0.0};
		const double sv[3] = {rev ? oxz[0] : z[0], rev ? oxz[1] : z[1], rev ? oxz[2] : z[2]};
		UNR for (int e = 0; e < 3; e++) {
			sm.zo[r][e] = sw[e];
			sm.zo[r][3 + e] = sv[e];
		}
		// ---- spatial inertia of this body about the world origin: m, h = m c, Io (xx yy zz xy xz yz)
		double Bd[10];
		{
			double c[3];
			oct_mat3_vec(Tw, cl, c);
			UNR for (int e = 0; e < 3; e++) c[e] += o[e];
			const double Il[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
			double RI[9], Iw[9];
			UNR for (int i = 0; i < 3; i++)
				UNR for (int j = 0; j < 3; j++) RI[3 * i + j] = Tw[3 * i] * Il[j] + Tw[3 * i + 1] * Il[3 + j] + Tw[3 * i + 2] * Il[6 + j];
			UNR for (int i = 0; i < 3; i++)
				UNR for (int j = 0; j < 3; j++) Iw[3 * i + j] = RI[3 * i] * Tw[3 * j] + RI[3 * i + 1] * Tw[3 * j + 1] + RI[3 * i + 2] * Tw[3 * j + 2];
			const double cc2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
			Bd[0] = mass;
			Bd[1] = mass * c[0];
			Bd[2] = mass * c[1];
			Bd[3] = mass * c[2];
			Bd[4] = Iw[0] + mass * (cc2 - c[0] * c[0]);
			Bd[5] = Iw[4] + mass * (cc2 - c[1] * c[1]);
			Bd[6] = Iw[8] + mass * (cc2 - c[2] * c[2]);
			Bd[7] = Iw[1] - mass * c[0] * c[1];
			Bd[8] = Iw[2] - mass * c[0] * c[2];
			Bd[9] = Iw[5] - mass * c[1] * c[2];
			if (!act) {
				UNR for (int e = 0; e < 10; e++) Bd[e] = 0.0;
			}
		}
		// I (w, v) = (L_O, p):  p = m v + w x h,  L_O = Io w + h x v
#define OCT_INERTIA_MUL(I, w, v, L, p)                                                                                     \
	{                                                                                                                      \
		const double h_[3] = {(I)[1], (I)[2], (I)[3]};                                                                     \
		const double wxh_[3] = OCT_CROSS(w, h_), hxv_[3] = OCT_CROSS(h_, v);                                               \
		(p)[0] = (I)[0] * (v)[0] + wxh_[0];                                                                                \
		(p)[1] = (I)[0] * (v)[1] + wxh_[1];                                                                                \
		(p)[2] = (I)[0] * (v)[2] + wxh_[2];                                                                                \
		(L)[0] = (I)[4] * (w)[0] + (I)[7] * (w)[1] + (I)[8] * (w)[2] + hxv_[0];                                            \
		(L)[1] = (I)[7] * (w)[0] + (I)[5] * (w)[1] + (I)[9] * (w)[2] + hxv_[1];                                            \
		(L)[2] = (I)[8] * (w)[0] + (I)[9] * (w)[1] + (I)[6] * (w)[2] + hxv_[2];                                            \
	}
		// ---- bias forces b(q, dq) + g(q)
		double V[6] = {sw[0] * dq_r, sw[1] * dq_r, sw[2] * dq_r, sv[0] * dq_r, sv[1] * dq_r, sv[2] * dq_r};
		if (!act) {
			UNR for (int e = 0; e < 6; e++) V[e] = 0.0;
		}
		oct_prefix_sum<6>(V, r);  // spatial velocity of body r about the world origin: (omega, v_O)
		double A[6];
		{
			// d/dt S_r = V_r x S_r (motion cross product): (w x s_w, w x s_v + v x s_w)
			const double wxsw[3] = OCT_CROSS(V, sw), wxsv[3] = OCT_CROSS(V, sv), vxsw[3] = OCT_CROSS(V + 3, sw);
			UNR for (int e = 0; e < 3; e++) {
				A[e] = act ? wxsw[e] * dq_r : 0.0;
				A[3 + e] = act ? (wxsv[e] + vxsw[e]) * dq_r : 0.0;
			}
		}
		oct_prefix_sum<6>(A, r);
		UNR for (int e = 0; e < 3; e++) A[3 + e] -= S.gravity[e];  // base acceleration -g: gravity as a fictitious acceleration of the whole chain
		double F[6];  // (moment about O, force)
		{
			double La[3], pa[3], Lv[3], pv[3];
			OCT_INERTIA_MUL(Bd, A, A + 3, La, pa)
			OCT_INERTIA_MUL(Bd, V, V + 3, Lv, pv)
			// V x* (L, p) = (w x L + v x p, w x p)
			const double wxL[3] = OCT_CROSS(V, Lv), vxp[3] = OCT_CROSS(V + 3, pv), wxp[3] = OCT_CROSS(V, pv);
			UNR for (int e = 0; e < 3; e++) {
				F[e] = La[e] + wxL[e] + vxp[e];
				F[3 + e] = pa[e] + wxp[e];
			}
		}
		oct_suffix_sum<6>(F, r);  // wrench transmitted through joint r
		const double h_r = sw[0] * F[0] + sw[1] * F[1] + sw[2] * F[2] + sv[0] * F[3] + sv[1] * F[4] + sv[2] * F[5];
		sm.rhs[r] = tau_r - h_r - S.damping * dq_r;
		// ---- M(q): composite inertias (suffix sums), row r of the lower triangle: M_ri = S_i . (Ic_r S_r), i <= r
		oct_suffix_sum<10>(Bd, r);
		oct_wave_sync();
		{
			double L[3], p[3];
			OCT_INERTIA_MUL(Bd, sw, sv, L, p)
			UNR for (int i = 0; i < N; i++) {
				double s = 0.0;
				UNR for (int e = 0; e < 3; e++) s = fma(sm.zo[i][e], L[e], fma(sm.zo[i][3 + e], p[e], s));
				sm.M[r][i] = s;
			}
		}
		oct_wave_sync();
		// ---- qdd = M^-1 rhs: Cholesky factor and both triangular solves in every lane
		{
			double L[N][N], dinv[N], x[N], y[N];
			UNR for (int i = 0; i < N; i++)
				UNR for (int j = 0; j <= i; j++) L[i][j] = sm.M[i][j];
			oct_cholesky<N>(L, dinv);
			UNR for (int i = 0; i < N; i++) {
				double s = sm.rhs[i];
				UNR for (int k = 0; k < i; k++) s = fma(-L[i][k], y[k], s);
				y[i] = s * dinv[i];
			}
			UNR for (int i = N - 1; i >= 0; i--) {
				double s = y[i];
				UNR for (int k = i + 1; k < N; k++) s = fma(-L[k][i], x[k], s);
				x[i] = s * dinv[i];
			}
			double mine = 0.0;
			UNR for (int i = 0; i < N; i++) mine = (i == rr) ? x[i] : mine;
			ddq_r = mine;
		}
		dq_r = fma(S.dt, ddq_r, dq_r);
		q_r = fma(S.dt, dq_r, q_r);
		oct_wave_sync();  // the next substep overwrites zo / rhs / M
	}
#undef OCT_INERTIA_MUL
	if (live && act) {
		S.q[(size_t)r * ld + b] = q_r;
		S.dq[(size_t)r * ld + b] = dq_r;
		if (S.ddq) S.ddq[(size_t)r * ld + b] = ddq_r;
	}
}


}  // namespace saip
