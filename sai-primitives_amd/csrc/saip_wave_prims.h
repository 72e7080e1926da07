// Register-resident dense linear algebra for ONE WAVEFRONT PER ROBOT INSTANCE (saip_kernel_wave.hip: chains of 9..32 dof).
//
// Every matrix of the instance lives in the wavefront's registers in the operand layout of v_mfma_f64_16x16x4_f64, so products need
// no data movement at all and nothing is synchronised by a barrier.  With lo = lane & 15, hi = lane >> 4:
//
//   "stored form" of a matrix X (4 KS rows, 16 T columns):   v[ks][t] = X[4 ks + hi][16 t + lo]          (one double per lane)
//
// This is at once the B operand of X (k index = hi), the A operand of X^T (row = lo, k = hi) and -- four consecutive ks at a time --
// the C/D tile of the instruction (row = 4 r + hi, column = lo; cdna_hip_programming.md:160-161).  Hence ONE product primitive,
//
//   wv_mm(S1, S2) = stored(S1^T S2),      contraction over the ROW index of both stored operands,
//
// whose result is again in stored form; every product of the control cycle is arranged to be of that shape (the transposed
// operand is obtained by swapping the roles of the two factors, never by moving data).  Symmetric matrices (M^-1, Lambda) are their
// own transposes.  Vectors live in LDS: "row distribution" x[4 ks + hi] feeds wv_matvec, which returns y = S^T x in "column
// distribution" y[16 t + lo] (the same value in the four lanes lo, lo + 16, lo + 32, lo + 48).
//
// SPD inverses are a blocked symmetric sweep (wv_sweep): pivot blocks of 4 (= the k depth of the instruction); the panel scaling
// W = P D^-1, the trailing update X -= W P^T and the fix-up of the pivot's block column all ride on the matrix cores; the 4 x 4
// diagonal block is inverted redundantly per lane from lane-broadcast (v_readlane) entries.  tools/proto/wave_forms.py is the NumPy
// model of this file's index algebra.
#pragma once
#include <hip/hip_runtime.h>

#include "saip_oct_common.h"

namespace saip {

typedef double wv_f64x4 __attribute__((ext_vector_type(4)));
typedef unsigned wv_u32x2 __attribute__((ext_vector_type(2)));

// ordering point for LDS traffic inside the (single) wavefront of a workgroup: LDS operations of one wavefront execute in order, so a
// wavefront-scope fence is all it takes (no s_barrier)
__device__ __forceinline__ void wv_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double wv_pack(const unsigned lo, const unsigned hi) { return __hiloint2double((int)hi, (int)lo); }

// sum over the four 16-lane rows (lanes lo, lo + 16, lo + 32, lo + 48), every lane gets the total, bitwise the same in all four:
// v_permlane16_swap / v_permlane32_swap (gfx950) exchange whole rows between two registers -- two VALU moves per dword, no LDS
__device__ __forceinline__ double wv_hsum(const double x) {
	const unsigned l = (unsigned)__double2loint(x), h = (unsigned)__double2hiint(x);
	const wv_u32x2 a = __builtin_amdgcn_permlane16_swap(l, l, false, false);
	const wv_u32x2 b = __builtin_amdgcn_permlane16_swap(h, h, false, false);
	const double y = wv_pack(a.x, b.x) + wv_pack(a.y, b.y);  // rows {0,1}: r0 + r1, rows {2,3}: r2 + r3
	const unsigned l2 = (unsigned)__double2loint(y), h2 = (unsigned)__double2hiint(y);
	const wv_u32x2 c = __builtin_amdgcn_permlane32_swap(l2, l2, false, false);
	const wv_u32x2 d = __builtin_amdgcn_permlane32_swap(h2, h2, false, false);
	return wv_pack(c.x, d.x) + wv_pack(c.y, d.y);
}

// value of the lane `n` places to the right inside the 16-lane row, cyclically (DPP row_ror)
template <int N>
__device__ __forceinline__ double wv_ror(const double x) {
	const int l = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x120 + N, 0xF, 0xF, false);
	const int h = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x120 + N, 0xF, 0xF, false);
	return __hiloint2double(h, l);
}
template <int N>
__device__ __forceinline__ int wv_ror_i(const int x) {
	return __builtin_amdgcn_update_dpp(0, x, 0x120 + N, 0xF, 0xF, false);
}
// value of the lane D places to the left (shr) / right (shl) inside the 16-lane row; lanes without such a neighbour get 0 -- or, with ONE, 1.0 (the
// diagonal of an identity frame): the high dword keeps its `old` operand 0x3FF00000 where DPP finds no source lane
template <int D, bool ONE>
__device__ __forceinline__ double wv_shr(const double x) {
	const int l = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x110 + D, 0xF, 0xF, true);
	const int h = ONE ? __builtin_amdgcn_update_dpp(0x3FF00000, __double2hiint(x), 0x110 + D, 0xF, 0xF, false)
					  : __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x110 + D, 0xF, 0xF, true);
	return __hiloint2double(h, l);
}
template <int D>
__device__ __forceinline__ double wv_shl(const double x) {
	const int l = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x100 + D, 0xF, 0xF, true);
	const int h = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x100 + D, 0xF, 0xF, true);
	return __hiloint2double(h, l);
}
// sum over the 16 lanes of a row; the pairing is commutative at every step, so all lanes end with bitwise the same value
__device__ __forceinline__ double wv_rowsum(double x) {
	x += wv_ror<8>(x);
	x += wv_ror<4>(x);
	x += wv_ror<2>(x);
	return x + wv_ror<1>(x);
}
__device__ __forceinline__ double wv_wavesum(const double x) { return wv_hsum(wv_rowsum(x)); }

// (value, index) arg-max over the 16 lanes of a row, ties to the lower index: a total order, so every lane ends with the same pair
#define WV_ARGMAX_STEP(N)                                                   \
	{                                                                       \
		const double ob = wv_ror<N>(best);                                  \
		const int oj = wv_ror_i<N>(jb);                                     \
		const bool take = ob > best || (ob == best && oj < jb);             \
		best = take ? ob : best;                                            \
		jb = take ? oj : jb;                                                \
	}
__device__ __forceinline__ void wv_row_argmax(double& best, int& jb) {
	WV_ARGMAX_STEP(8)
	WV_ARGMAX_STEP(4)
	WV_ARGMAX_STEP(2)
	WV_ARGMAX_STEP(1)
}

// A value every lane holds identically, re-read through the scalar unit: tells the compiler it is wave-uniform, so that the branches it
// feeds become scalar branches instead of EXEC-masked regions (an instance's whole control flow is uniform: one wavefront, one instance)
__device__ __forceinline__ double wv_uniform(const double x) {
	const int l = __builtin_amdgcn_readfirstlane(__double2loint(x));
	const int h = __builtin_amdgcn_readfirstlane(__double2hiint(x));
	return __hiloint2double(h, l);
}
__device__ __forceinline__ bool wv_uniform(const bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }

__device__ __forceinline__ double wv_readlane(const double x, const int src) {  // src wave-uniform
	const int l = __builtin_amdgcn_readlane(__double2loint(x), src);
	const int h = __builtin_amdgcn_readlane(__double2hiint(x), src);
	return __hiloint2double(h, l);
}

// C (+)= sign * S1^T S2.  A: stored S1 (4 KS x 16 TA), B: stored S2 (4 KS x 16 TB); C: the first RK register rows of the stored
// 16 TA x 16 TB result (RK = 4 TA keeps all of it; a k x n result with k <= 8 keeps RK = 2).  The TA x TB tiles are independent
// accumulation chains, interleaved over ks.
template <bool ACC, bool NEG, int KS, int TA, int TB, int RK>
__device__ __forceinline__ void wv_mm(const double (&A)[KS][TA], const double (&B)[KS][TB], double (&C)[RK][TB]) {
	static_assert(RK <= 4 * TA, "result rows");
	wv_f64x4 acc[TA][TB];
	UNR for (int rt = 0; rt < TA; rt++)
		UNR for (int ct = 0; ct < TB; ct++)
			UNR for (int r = 0; r < 4; r++) acc[rt][ct][r] = (ACC && 4 * rt + r < RK) ? C[4 * rt + r][ct] : 0.0;
	UNR for (int ks = 0; ks < KS; ks++)
		UNR for (int rt = 0; rt < TA; rt++) {
			if (4 * rt >= RK) continue;
			const double a = NEG ? -A[ks][rt] : A[ks][rt];
			UNR for (int ct = 0; ct < TB; ct++) acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, B[ks][ct], acc[rt][ct], 0, 0, 0);
		}
	UNR for (int rt = 0; rt < TA; rt++)
		UNR for (int ct = 0; ct < TB; ct++)
			UNR for (int r = 0; r < 4; r++)
				if (4 * rt + r < RK) C[4 * rt + r][ct] = acc[rt][ct][r];
}

// y = S^T x in column distribution (y[t] = entry 16 t + lo, the same in all four rows); xr = x in row distribution (xr[ks] = x[4 ks + hi])
template <int KS, int T>
__device__ __forceinline__ void wv_matvec(const double (&S)[KS][T], const double (&xr)[KS], double (&y)[T]) {
	UNR for (int t = 0; t < T; t++) {
		double s = 0.0;
		UNR for (int ks = 0; ks < KS; ks++) s = fma(S[ks][t], xr[ks], s);
		y[t] = wv_hsum(s);
	}
}
template <int KS>
__device__ __forceinline__ void wv_load_rows(const double* x, const int hi, double (&xr)[KS]) {
	UNR for (int ks = 0; ks < KS; ks++) xr[ks] = x[4 * ks + hi];
}

// In place: stored symmetric positive definite X (16 T x 16 T) -> stored -X^-1 on the leading 4 nb rows / columns (nb wave-uniform;
// what lies beyond must be the identity -- it is left alone).  Returns the smallest LDL^T pivot met: X is positive definite iff it is
// positive (the singular-value certificates test it against their tolerance; a caller that gets a non-positive one discards the
// result; a non-positive 2 x 2 determinant shows as a non-positive pivot).  Block step kb (pivot rows 4 kb .. 4 kb + 3, living in column tile t0 = kb / 4 at lanes lo = rho .. rho + 3, rho = 4 (kb % 4)):
//   P_t = v[kb][t]                       the pivot's block row = (by symmetry) its block column, already an MFMA operand
//   D   = the 4 x 4 diagonal block       16 lanes of P_t0; rebuilt from v_readlane broadcasts (scalar registers) and inverted by 2 x 2 blocks in closed form
//   W_t = P_t D^-1                       one MFMA per tile: A operand = D^-1 on rows rho .., B operand = P_t; lands in register kb % 4
//   X  -= W P'^T                         T x T MFMAs; P' = P with -I on the pivot's own columns and those columns of X zeroed first, so
//                                        that the block column comes out as +W exactly (sweep operator: a_ik <- a_ik / a_kk)
//   v[kb][t] = W_t, diagonal block = -D^-1
// PIVOTS_LAST: the caller only wants the pivots (a positive-definiteness certificate): the last block step stops behind its four pivots -- no
// inverse of that block, no panel, no update (v is then NOT an inverse).
template <int T, bool PIVOTS_LAST = false>
__device__ __forceinline__ double wv_sweep(double (&v)[4 * T][T], const int nb, const int lo, const int hi) {
	double minpiv = 1.0e300;
	bool nanpiv = false;
	// which entry of the (symmetric) 4 x 4 inverse a lane of a pivot block takes: D^-1[a][hi] with a = lo & 3 -- the same in every block step of the
	// sweep, so the ten selection masks are formed once here (they used to be rebuilt from a = lo - rho in each of the 4 T steps)
	const int sa = lo & 3, smax = sa > hi ? sa : hi, smin = sa > hi ? hi : sa;
	const int scode = smax * 4 + smin;
	UNR for (int kb = 0; kb < 4 * T; kb++) {
		if (kb < nb) {  // wave-uniform
			const int t0 = kb >> 2, rho = 4 * (kb & 3);
			double P[T];
			UNR for (int t = 0; t < T; t++) P[t] = v[kb][t];
			// D[b][a] sits in lane (rho + a) + 16 b of P[t0]; lower triangle by broadcast (wave-uniform values: scalar registers)
			double D[4][4];
			UNR for (int i = 0; i < 4; i++)
				UNR for (int j2 = 0; j2 <= i; j2++) D[i][j2] = wv_readlane(P[t0], (rho + j2) + 16 * i);
			// D^-1 by 2 x 2 blocks in closed form, D = [A B^T; B C]:  A^-1 = adj(A) / det A,  E = B A^-1,  S = C - E B^T,  S^-1 = adj(S) / det S,
			//   D^-1 = [A^-1 + E^T S^-1 E, -(S^-1 E)^T; -S^-1 E, S^-1].
			// Two reciprocals in sequence where the 4 x 4 Cholesky factor + substitution this replaces had four reciprocal square roots and two
			// triangular solves (round 4: the dependent chain of a pivot block ~650 -> ~250 clocks; 38 pivot blocks per config-5 instance).  The LDL^T
			// pivots the certificates look at are d00, det A / d00, s00, det S / s00 (their reciprocals are off the chain: hardware estimate).
			double Di[4][4];  // lower triangle of D^-1
			{
				const double a00 = D[0][0], a10 = D[1][0], a11 = D[1][1];
				const double b00 = D[2][0], b01 = D[2][1], b10 = D[3][0], b11 = D[3][1];
				const double c00 = D[2][2], c10 = D[3][2], c11 = D[3][3];
				const double detA = fma(a00, a11, -a10 * a10);
				const double iA = oct_rcp(detA);
				const double ai00 = a11 * iA, ai10 = -a10 * iA, ai11 = a00 * iA;
				const double e00 = fma(b00, ai00, b01 * ai10), e01 = fma(b00, ai10, b01 * ai11);
				const double e10 = fma(b10, ai00, b11 * ai10), e11 = fma(b10, ai10, b11 * ai11);
				const double s00 = c00 - fma(e00, b00, e01 * b01), s10 = c10 - fma(e10, b00, e11 * b01), s11 = c11 - fma(e10, b10, e11 * b11);
				const double detS = fma(s00, s11, -s10 * s10);
				const double iS = oct_rcp(detS);
				const double si00 = s11 * iS, si10 = -s10 * iS, si11 = s00 * iS;
				if (PIVOTS_LAST && kb == nb - 1) {
					const double p1 = detA * __builtin_amdgcn_rcp(a00), p3 = detS * __builtin_amdgcn_rcp(s00);
					minpiv = fmin(fmin(minpiv, a00), fmin(p1, fmin(s00, p3)));
					const double psum = (a00 + p1) + (s00 + p3);
					nanpiv = nanpiv || (psum != psum);
					continue;
				}
				const double f00 = fma(si00, e00, si10 * e10), f01 = fma(si00, e01, si10 * e11);
				const double f10 = fma(si10, e00, si11 * e10), f11 = fma(si10, e01, si11 * e11);
				Di[0][0] = fma(e00, f00, fma(e10, f10, ai00));
				Di[1][0] = fma(e01, f00, fma(e11, f10, ai10));
				Di[1][1] = fma(e01, f01, fma(e11, f11, ai11));
				Di[2][0] = -f00;
				Di[2][1] = -f01;
				Di[3][0] = -f10;
				Di[3][1] = -f11;
				Di[2][2] = si00;
				Di[3][2] = si10;
				Di[3][3] = si11;
				// smallest pivot so far (v_min ignores a NaN operand, so NaNs are tracked on the side: a NaN anywhere in the block shows in the sum)
				const double p1 = detA * __builtin_amdgcn_rcp(a00), p3 = detS * __builtin_amdgcn_rcp(s00);
				minpiv = fmin(fmin(minpiv, a00), fmin(p1, fmin(s00, p3)));
				const double psum = (a00 + p1) + (s00 + p3);
				nanpiv = nanpiv || (psum != psum);
			}
			const int a = lo - rho;
			const bool inblk = (lo >> 2) == (kb & 3);
			// D^-1[a][hi] on the pivot's lanes, 0 elsewhere (symmetric: entry (max, min) of the lower triangle)
			double dl = Di[0][0];
			UNR for (int i = 1; i < 4; i++)
				UNR for (int j2 = 0; j2 <= i; j2++) dl = (scode == 4 * i + j2) ? Di[i][j2] : dl;
			dl = inblk ? dl : 0.0;
			double W[T];
			UNR for (int t = 0; t < T; t++) {
				wv_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
				acc = __builtin_amdgcn_mfma_f64_16x16x4f64(dl, P[t], acc, 0, 0, 0);
				W[t] = acc[kb & 3];  // W[16 t + lo][hi]
			}
			const double keep = inblk ? 0.0 : 1.0;
			UNR for (int ks = 0; ks < 4 * T; ks++) v[ks][t0] *= keep;  // (one multiplication per entry where a select costs two moves; a NaN or infinity here is a failed sweep anyway)
			double Pm[T];
			UNR for (int t = 0; t < T; t++) Pm[t] = P[t];
			Pm[t0] = inblk ? ((hi == a) ? -1.0 : 0.0) : Pm[t0];
			UNR for (int rt = 0; rt < T; rt++)
				UNR for (int ct = 0; ct < T; ct++) {
					wv_f64x4 acc;
					UNR for (int r = 0; r < 4; r++) acc[r] = v[4 * rt + r][ct];
					acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-W[rt], Pm[ct], acc, 0, 0, 0);
					UNR for (int r = 0; r < 4; r++) v[4 * rt + r][ct] = acc[r];
				}
			UNR for (int t = 0; t < T; t++) v[kb][t] = W[t];
			v[kb][t0] = inblk ? -dl : v[kb][t0];
		}
	}
	const double mp = wv_uniform(minpiv);
	return wv_uniform(nanpiv) ? __builtin_nan("") : mp;  // (a NaN pivot fails every "> tolerance" test of the callers)
}

template <int KS, int T>
__device__ __forceinline__ void wv_negate(double (&v)[KS][T]) {
	UNR for (int ks = 0; ks < KS; ks++)
		UNR for (int t = 0; t < T; t++) v[ks][t] = -v[ks][t];
}
// identity on the diagonal entries i >= from (i < upto), untouched elsewhere: pads a k x k block to the next multiple of the pivot block
template <int KS, int T>
__device__ __forceinline__ void wv_pad_identity(double (&v)[KS][T], const int from, const int upto, const int lo, const int hi) {
	UNR for (int ks = 0; ks < KS; ks++)
		UNR for (int t = 0; t < T; t++) {
			const int i = 4 * ks + hi, j = 16 * t + lo;
			const bool diag = (i == j), pad = (i >= from || j >= from) && i < upto && j < upto;
			v[ks][t] = pad ? (diag ? 1.0 : 0.0) : v[ks][t];
		}
}

}  // namespace saip
