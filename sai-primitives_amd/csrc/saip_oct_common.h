// Building blocks shared by the eight-lanes-per-instance kernels (saip_kernel_oct.hip, saip_dynamics_oct.hip): lane r of an 8-lane group
// owns joint r / row r of its instance; two instances share a 16-lane DPP row.
#pragma once
#include <hip/hip_runtime.h>

namespace saip {

#define UNR _Pragma("unroll")

// a load through a pointer that was itself read from memory (TaskDev::goal, ::integ): the compiler only knows it as a generic address and emits
// flat_load, which may complete out of order with the global loads around it -- every wait behind it then becomes a wait for EVERYTHING in
// flight (vmcnt(0)).  The arrays are in HBM: say so, and the waits stay counted.
__device__ __forceinline__ double saip_ldg(const double* p, const size_t i) {
	typedef const double __attribute__((address_space(1))) * gptr_t;
	return ((gptr_t)(unsigned long long)p)[i];
}

// 1/x and 1/sqrt(x) from the hardware estimates plus two Newton steps: full double accuracy for the normal-range, positive arguments met
// here (pivots of SPD factorisations), without the scaling / fix-up sequences of the IEEE division and the library rsqrt
__device__ __forceinline__ double oct_rcp(const double x) {
	double y = __builtin_amdgcn_rcp(x);
#if !defined(SAIP_OCT_EXP_NEWTON1)
	y = fma(fma(-x, y, 1.0), y, y);
#endif
	return fma(fma(-x, y, 1.0), y, y);
}
__device__ __forceinline__ double oct_rsqrt(const double x) {
	double y = __builtin_amdgcn_rsq(x);
	const double h = 0.5 * x;
#if !defined(SAIP_OCT_EXP_NEWTON1)
	y = fma(y, fma(-h * y, y, 0.5), y);
#endif
	return fma(y, fma(-h * y, y, 0.5), y);
}

// value of the lane `n` positions up (row_shl) / down (row_shr) inside the 16-lane DPP row, 0 beyond it: a VALU move, no LDS round trip.
// Callers mask the lanes whose source sits in the neighbouring instance (two instances share a row).
template <int CTRL>
__device__ __forceinline__ double oct_dpp(const double x) {
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
	return __hiloint2double(hi, lo);
}
enum { DPP_ROW_SHL = 0x100, DPP_ROW_SHR = 0x110, DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141 };
// sum / maximum over the eight lanes of an instance, every lane gets the same result: pairs and quads by quad permutes, then the
// other quad of the 8-lane half-row by row_half_mirror (lane i <-> 7 - i)
__device__ __forceinline__ double oct_group_sum(double x) {
	x += oct_dpp<DPP_QUAD_XOR1>(x);
	x += oct_dpp<DPP_QUAD_XOR2>(x);
	return x + oct_dpp<DPP_HALF_MIRROR>(x);
}
__device__ __forceinline__ double oct_group_max(double x) {
	x = fmax(x, oct_dpp<DPP_QUAD_XOR1>(x));
	x = fmax(x, oct_dpp<DPP_QUAD_XOR2>(x));
	return fmax(x, oct_dpp<DPP_HALF_MIRROR>(x));
}

// Which block of instances a workgroup takes.  The dispatcher is observed to deal workgroups round-robin over the 8 XCDs (workgroups w and w + 8
// share one; MI355X_MICROARCH.md "Workgroup dispatch, XCD placement"), each with a private L2.  With block = blockIdx, neighbouring blocks -- which
// share cache lines of every [row][ld] input array (8 instances x 8 B = half a 128-B line per row for the eight-lane kernels, ONE double per row for
// the wavefront-per-instance kernel) -- sit on different XCDs, and every line is fetched from HBM once per XCD that touches it (measured, round 3 / 4:
// 2.15 x the algorithmic bytes on config 2, 8 x on config 5).  This bijective remap gives the workgroups of one XCD a contiguous range of blocks.
// A speed choice only: any other placement reads the same values.  (SAIP_NO_XCD_REMAP: identity, for A/B builds.)
// Used by the wavefront-per-instance kernel only (config 5, same session: 172.7 -> 170.2 us per launch, HBM traffic 65 -> 11 MB against 7.5 MB
// algorithmic).  The eight-lane kernels were measured with it as well and run on plain blockIdx: their traffic fell as expected (config 2: 4.6 ->
// 2.7 MB per launch against 2.2 algorithmic) but every launch got SLOWER (config 2 6.67 -> 6.92 us, config 3 13.53 -> 13.85, config 4's shard 9.61 ->
// 10.02; tools/ab_remap.sh) -- at 4 % of the HBM peak the bytes are not what these kernels wait for, and neighbouring workgroups that miss on the
// same line at the same moment queue behind one another in one L2 instead of missing in parallel in two.
__device__ __forceinline__ int saip_xcd_block(const unsigned orig, const unsigned nwg) {
#if defined(SAIP_NO_XCD_REMAP)
	return (int)orig;
#else
	const unsigned q = nwg >> 3, r = nwg & 7u, x = orig & 7u;
	return (int)((x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + (orig >> 3));
#endif
}

// ---------------------------------------------------------------- lane layout
// Eight lanes per instance, eight instances per wavefront, two instances per 16-lane DPP row.  INTERLEAVED (default): the two instances of a
// row take its even and its odd lanes (joint r of instance 2 row + parity sits in lane 16 row + 2 r + parity).  Every boundary of a scan along
// the chain is then a ROW boundary, which DPP handles by itself (bound_ctrl zero fill, or the untouched `old` operand for an identity) --
// with the two instances side by side (lanes 0-7 / 8-15, SAIP_OCT_ILV=0) every scan step needed a select per moved register: a third of the
// instructions of the wavefront's longest phase (doubling them cost 4.5 % of the launch, measured).
#ifndef SAIP_OCT_ILV
#define SAIP_OCT_ILV 1
#endif
constexpr bool OCT_ILV = SAIP_OCT_ILV != 0;
__device__ __forceinline__ int octl_r(const int lane) { return OCT_ILV ? ((lane >> 1) & 7) : (lane & 7); }
__device__ __forceinline__ int octl_grp(const int lane) { return OCT_ILV ? (((lane >> 4) << 1) | (lane & 1)) : (lane >> 3); }
// the lane that holds joint j of this lane's instance
__device__ __forceinline__ int octl_src(const int lane, const int j) { return OCT_ILV ? ((lane & 0x31) | (j << 1)) : ((lane & ~7) | j); }
// the votes of the eight lanes of this lane's instance, bit r = joint r
__device__ __forceinline__ unsigned octl_votes(const unsigned long long ballot, const int lane) {
	if (!OCT_ILV) return (unsigned)((ballot >> ((lane >> 3) * 8)) & 0xffull);
	unsigned x = (unsigned)((ballot >> (16 * (lane >> 4))) & 0xffffull) >> (lane & 1);  // the row's votes, this instance's on the even bits
	x &= 0x5555u;
	x = (x | (x >> 1)) & 0x3333u;
	x = (x | (x >> 2)) & 0x0f0fu;
	x = (x | (x >> 4)) & 0x00ffu;
	return x;
}
enum { DPP_ROW_ROR = 0x120 };
// sum / maximum over the eight lanes of an instance, every lane gets the result (rotations by 4 and 8 inside the row keep the parity)
__device__ __forceinline__ double octl_sum(double x) {
	if (!OCT_ILV) return oct_group_sum(x);
	x += oct_dpp<DPP_QUAD_XOR2>(x);
	x += oct_dpp<DPP_ROW_ROR + 4>(x);
	return x + oct_dpp<DPP_ROW_ROR + 8>(x);
}
__device__ __forceinline__ double octl_max(double x) {
	if (!OCT_ILV) return oct_group_max(x);
	x = fmax(x, oct_dpp<DPP_QUAD_XOR2>(x));
	x = fmax(x, oct_dpp<DPP_ROW_ROR + 4>(x));
	return fmax(x, oct_dpp<DPP_ROW_ROR + 8>(x));
}
// the three exchange steps of a butterfly over the eight lanes of an instance (arg-max with its index, ...)
constexpr int OCT_BFLY0 = OCT_ILV ? (int)DPP_QUAD_XOR2 : (int)DPP_QUAD_XOR1, OCT_BFLY1 = OCT_ILV ? DPP_ROW_ROR + 4 : (int)DPP_QUAD_XOR2,
			  OCT_BFLY2 = OCT_ILV ? DPP_ROW_ROR + 8 : (int)DPP_HALF_MIRROR;
// value of joint r - D (shr) / r + D (shl) of the same instance; lanes without such a joint get `fill` (0 or 1: the identity's diagonal).
// Interleaved layout only (the side-by-side layout masks by selects at the call sites).
template <int D, bool ONE>
__device__ __forceinline__ double octl_shr(const double x) {
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), DPP_ROW_SHR + 2 * D, 0xF, 0xF, true);
	const int hi = ONE ? __builtin_amdgcn_update_dpp(0x3FF00000, __double2hiint(x), DPP_ROW_SHR + 2 * D, 0xF, 0xF, false)
					   : __builtin_amdgcn_update_dpp(0, __double2hiint(x), DPP_ROW_SHR + 2 * D, 0xF, 0xF, true);
	return __hiloint2double(hi, lo);
}
template <int D>
__device__ __forceinline__ double octl_shl(const double x) {
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), DPP_ROW_SHL + 2 * D, 0xF, 0xF, true);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), DPP_ROW_SHL + 2 * D, 0xF, 0xF, true);
	return __hiloint2double(hi, lo);
}
// one step of the prefix product of the frames along the chain: T_r <- T_{r-D} o T_r, (Ra,oa) o (Rb,ob) = (Ra Rb, oa + Ra ob)
#define OCT_FRAME_STEP(D, Tw, r)                                                                                                     \
	{                                                                                                                                \
		double Xa[12], Tn[12];                                                                                                       \
		if (OCT_ILV) {                                                                                                               \
			UNR for (int e = 0; e < 12; e++) Xa[e] = (e == 0 || e == 4 || e == 8) ? octl_shr<D, true>(Tw[e]) : octl_shr<D, false>(Tw[e]); \
		} else {                                                                                                                     \
			UNR for (int e = 0; e < 12; e++) Xa[e] = oct_dpp<DPP_ROW_SHR + D>(Tw[e]);                                                \
		}                                                                                                                            \
		UNR for (int i = 0; i < 3; i++) {                                                                                            \
			UNR for (int j = 0; j < 3; j++) Tn[3 * i + j] = Xa[3 * i] * Tw[j] + Xa[3 * i + 1] * Tw[3 + j] + Xa[3 * i + 2] * Tw[6 + j];  \
			Tn[9 + i] = Xa[9 + i] + Xa[3 * i] * Tw[9] + Xa[3 * i + 1] * Tw[10] + Xa[3 * i + 2] * Tw[11];                            \
		}                                                                                                                            \
		const bool take = OCT_ILV || r >= D;  /* (interleaved: lanes without a predecessor composed with the identity) */            \
		UNR for (int e = 0; e < 12; e++) Tw[e] = take ? Tn[e] : Tw[e];                                                               \
	}


__device__ __forceinline__ void oct_mat3_vec(const double* R, const double* v, double* out) {
	UNR for (int i = 0; i < 3; i++) out[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}

// Cholesky factor of an SPD R x R matrix given by its lower triangle; L overwrites A's lower triangle, dinv = 1 / L_kk
template <int R>
__device__ __forceinline__ void oct_cholesky(double (&L)[R][R], double (&dinv)[R]) {
	UNR for (int k = 0; k < R; k++) {
		const double rs = oct_rsqrt(L[k][k]);
		dinv[k] = rs;
		UNR for (int i = k + 1; i < R; i++) L[i][k] *= rs;
		UNR for (int j = k + 1; j < R; j++)
			UNR for (int i = j; i < R; i++) L[i][j] = fma(-L[i][k], L[j][k], L[i][j]);
	}
}
// x = (L L^T)^-1 e_col, i.e. column (= row) `col` of the inverse; col is a lane variable
template <int R>
__device__ __forceinline__ void oct_solve_unit(const double (&L)[R][R], const double (&dinv)[R], const int col, double (&x)[R]) {
	double y[R];
	UNR for (int i = 0; i < R; i++) {
		double s = (i == col) ? 1.0 : 0.0;
		UNR for (int k = 0; k < i; k++) s = fma(-L[i][k], y[k], s);
		y[i] = s * dinv[i];
	}
	UNR for (int i = R - 1; i >= 0; i--) {
		double s = y[i];
		UNR for (int k = i + 1; k < R; k++) s = fma(-L[k][i], x[k], s);
		x[i] = s * dinv[i];
	}
}

// x = (L L^T)^-1 b for any right-hand side held by the lane
template <int R>
__device__ __forceinline__ void oct_solve(const double (&L)[R][R], const double (&dinv)[R], const double (&b)[R], double (&x)[R]) {
	double y[R];
	UNR for (int i = 0; i < R; i++) {
		double s = b[i];
		UNR for (int k = 0; k < i; k++) s = fma(-L[i][k], y[k], s);
		y[i] = s * dinv[i];
	}
	UNR for (int i = R - 1; i >= 0; i--) {
		double s = y[i];
		UNR for (int k = i + 1; k < R; k++) s = fma(-L[k][i], x[k], s);
		x[i] = s * dinv[i];
	}
}

// inverse of an SPD R x R matrix given by its lower triangle
template <int R>
__device__ __forceinline__ void oct_spd_inverse_n(const double (&A)[R][R], double (&Ainv)[R][R]) {
	double L[R][R], dinv[R], x[R];
	UNR for (int i = 0; i < R; i++)
		UNR for (int j = 0; j <= i; j++) L[i][j] = A[i][j];
	oct_cholesky<R>(L, dinv);
	UNR for (int col = 0; col < R; col++) {
		oct_solve_unit<R>(L, dinv, col, x);
		UNR for (int i = 0; i < R; i++) Ainv[i][col] = x[i];
	}
}

// cyclic Jacobi eigen-solve of a symmetric n x n matrix in LDS by ONE lane (rare exact path): eigenvalues on the diagonal of A,
// eigenvectors in the columns of V (optional)
__device__ inline void oct_serial_jacobi(double* A, double* V, const int n, const int ld) {
	if (V)
		for (int i = 0; i < n; i++)
			for (int j = 0; j < n; j++) V[i * ld + j] = (i == j) ? 1.0 : 0.0;
	for (int sweep = 0; sweep < 14; sweep++) {
		double off = 0.0, dg = 0.0;
		for (int p = 0; p < n; p++) {
			dg = fma(A[p * ld + p], A[p * ld + p], dg);
			for (int q = p + 1; q < n; q++) off = fma(A[p * ld + q], A[p * ld + q], off);
		}
		if (!(off > 1e-36 * dg)) break;
		for (int p = 0; p < n - 1; p++)
			for (int q = p + 1; q < n; q++) {
				const double apq = A[p * ld + q];
				if (apq == 0.0) continue;
				const double theta = (A[q * ld + q] - A[p * ld + p]) / (2.0 * apq);
				const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(fma(theta, theta, 1.0)));
				const double c = rsqrt(fma(t, t, 1.0)), s = t * c;
				for (int k = 0; k < n; k++) {  // A <- A R (columns p, q)
					const double akp = A[k * ld + p], akq = A[k * ld + q];
					A[k * ld + p] = c * akp - s * akq;
					A[k * ld + q] = s * akp + c * akq;
				}
				for (int k = 0; k < n; k++) {  // A <- R^T A (rows p, q)
					const double apk = A[p * ld + k], aqk = A[q * ld + k];
					A[p * ld + k] = c * apk - s * aqk;
					A[q * ld + k] = s * apk + c * aqk;
				}
				if (V)
					for (int k = 0; k < n; k++) {
						const double vkp = V[k * ld + p], vkq = V[k * ld + q];
						V[k * ld + p] = c * vkp - s * vkq;
						V[k * ld + q] = s * vkp + c * vkq;
					}
			}
	}
}

}  // namespace saip
