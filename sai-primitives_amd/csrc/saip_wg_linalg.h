// Workgroup-cooperative dense linear algebra on LDS-resident matrices (FP64), shared by the cycle kernels.
// T = threads of the workgroup; every routine is entered by all T threads and ends with all results visible.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace saip {

// the synchronisation point between two steps of a routine: a workgroup barrier -- or, where the "workgroup" is ONE wavefront inside a larger
// kernel (the slow tail of the eight-lane kernels: T = 64, the other wavefront of the workgroup has left), whatever the includer defines:
// LDS operations of one wavefront execute in order, a wavefront-scope fence is enough there
#if defined(SAIP_WG_SYNC)
#define SYNC() SAIP_WG_SYNC()
#else
#define SYNC() __syncthreads()
#endif

enum { GEMM_SET = 0, GEMM_I_MINUS = 1, GEMM_ADD = 2, GEMM_SUB = 3 };

// 1/x for the pivots of SPD factorisations (positive, normal range): hardware estimate + two Newton steps, ~9 instructions instead of the
// ~35 of the IEEE division sequence (every thread of the workgroup computes it at every pivot of the sweep inverse)
__device__ __forceinline__ double wg_rcp(const double x) {
	double y = __builtin_amdgcn_rcp(x);
	y = fma(fma(-x, y, 1.0), y, y);
	return fma(fma(-x, y, 1.0), y, y);
}
// 1/sqrt(x), x > 0 in the normal range: the same for the reciprocal square root
__device__ __forceinline__ double wg_rsqrt(const double x) {
	double y = __builtin_amdgcn_rsq(x);
	const double h = 0.5 * x;
	y = fma(y, fma(-h * y, y, 0.5), y);
	return fma(y, fma(-h * y, y, 0.5), y);
}

typedef double saip_f64x4 __attribute__((ext_vector_type(4)));

// C(m x p) (op)= A(m x k) * B(k x p); A(i,l) = A[i*ars + l*acs], B(l,j) = B[l*brs + j*bcs]  (strides express transposes)
// Four-wavefront workgroups (dof > 8) run products with at least 128 outputs on the FP64 matrix cores: the <= 32 x 32 result is
// four 16 x 16 tiles, one per wavefront, accumulated by v_mfma_f64_16x16x4_f64 over k in steps of 4 (operands straight from LDS,
// zero-filled outside the matrix).  f64 MFMA layout (cdna_hip_programming.md:160-161): A[l&15][l>>4], B[l>>4][l&15], D[(l>>4)+4r][l&15].
template <int T>
__device__ __forceinline__ void gemm(double* C, int ldc, const double* A, int ars, int acs, const double* Bm, int brs, int bcs,
									 int m, int k, int p, int tid, int mode = GEMM_SET) {
	if (T >= 256 && (m * p >= 128 || k >= 8)) {  // (small results with a long inner dimension, e.g. A = T1 J^T: one tile beats k serial LDS round trips per output)
		const int wave = tid >> 6, lane = tid & 63;
		const int r0 = 16 * (wave >> 1), c0 = 16 * (wave & 1);
		if (wave < 4 && r0 < m && c0 < p) {  // wave-uniform
			saip_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
			const int ar = r0 + (lane & 15), bc = c0 + (lane & 15), kq = lane >> 4;
			for (int kk = 0; kk < k; kk += 4) {
				const int kx = kk + kq;
				const double a = (ar < m && kx < k) ? A[ar * ars + kx * acs] : 0.0;
				const double b = (kx < k && bc < p) ? Bm[kx * brs + bc * bcs] : 0.0;
				acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
			}
			for (int r = 0; r < 4; r++) {
				const int i = r0 + (lane >> 4) + 4 * r, j = bc;
				if (i < m && j < p) {
					const double s = acc[r];
					double* c = C + i * ldc + j;
					if (mode == GEMM_SET) *c = s;
					else if (mode == GEMM_I_MINUS) *c = (i == j ? 1.0 : 0.0) - s;
					else if (mode == GEMM_ADD) *c += s;
					else *c -= s;
				}
			}
		}
		return;
	}
	for (int e = tid; e < m * p; e += T) {
		int i = e / p, j = e - i * p;
		double s = 0.0;
		for (int l = 0; l < k; l++) s = fma(A[i * ars + l * acs], Bm[l * brs + j * bcs], s);
		double* c = C + i * ldc + j;
		if (mode == GEMM_SET) *c = s;
		else if (mode == GEMM_I_MINUS) *c = (i == j ? 1.0 : 0.0) - s;
		else if (mode == GEMM_ADD) *c += s;
		else *c -= s;
	}
}

// Aout = Ain^-1 for a symmetric positive definite n x n matrix.  WL, WX: n x n work (ld); none of the four may alias.
// Symmetric sweep operator: sweeping pivot k maps  a_kk -> -1/a_kk,  a_ik -> a_ik/a_kk,  a_ij -> a_ij - a_ik a_kj / a_kk;
// after all n pivots (each a positive Schur-complement diagonal) the matrix is -A^-1.  Every step is one full-matrix
// rank-one update by all T threads reading the previous buffer and writing the other one: ONE barrier per pivot, no
// serial triangular solves.  The product a_ik*a_kj is formed first so that (i,j) and (j,i) stay bitwise equal.
// n <= 8 for the one-wavefront variant, n <= 32 for the four-wavefront variant: at most EPT elements per thread, whose (i, j) are
// fixed across pivots and computed once.
template <int T>
__device__ void spd_inverse(const double* Ain, double* Aout, double* WL, double* WX, double* /*dv*/, int n, int ld, int tid) {
	constexpr int EPT = (T == 64) ? 1 : (T == 256 ? 4 : 2);
	int ei[EPT], ej[EPT];
	for (int r = 0; r < EPT; r++) {
		const int e = tid + r * T;
		ei[r] = e < n * n ? e / n : -1;
		ej[r] = e - ei[r] * n;
	}
	const double* src = Ain;
	for (int k = 0; k < n; k++) {
		const bool last = (k == n - 1);
		double* dst = last ? Aout : ((k & 1) ? WX : WL);
		const double sgn = last ? -1.0 : 1.0;
		const double rd = wg_rcp(src[k * ld + k]);
#pragma unroll
		for (int r = 0; r < EPT; r++) {
			const int i = ei[r], j = ej[r];
			if (i < 0) continue;
			const double aik = src[i * ld + k], akj = src[k * ld + j], aij = src[i * ld + j];
			double v = fma(-(aik * akj), rd, aij);
			if (i == k) v = (j == k) ? -rd : akj * rd;
			else if (j == k) v = aik * rd;
			dst[i * ld + j] = sgn * v;
		}
		SYNC();
		src = dst;
	}
}

// M_BIE^-1 from M^-1 when only a few diagonal entries of M are clamped (SingularityHandler.cpp:176-182: M_BIE = M + sum_e c_e e e^T,
// c_e = thr - M_ee > 0): Woodbury, M_BIE^-1 = M^-1 - Y K^-1 Y^T with Y = the clamped columns of M^-1 and K = C^-1 + Y[idx, :] (c x c, SPD).
// One thread inverts K (c <= 8); everything else is two small passes over the matrix -- five barriers instead of the n pivots of a
// second full inverse (30-dof chain: 3 clamped entries, 37 k clocks -> 3 k).  Returns false (nothing written) when more than 8 entries
// are clamped: the caller then inverts M_BIE in full.  K: >= 64 doubles, idx: >= 9 ints, Z: n x 8 work (ld), all in LDS.
template <int T>
__device__ bool bie_inverse_lowrank(const double* Mm, const double* Minv, double* MBinv, double* K, int* idx, double* Z, const double thr,
									 int n, int ld, int tid) {
	if (tid < 64) {  // the first wavefront: lane i looks at diagonal entry i (n <= 32), positions by a ballot prefix count (deterministic order)
		const bool cl = tid < n && Mm[tid * ld + tid] < thr;
		const unsigned long long mask = __ballot(cl);
		const int pos = __popcll(mask & ((1ull << tid) - 1ull));
		if (cl && pos < 8) idx[pos] = tid;
		if (tid == 0) idx[8] = __popcll(mask);
	}
	SYNC();
	const int c = idx[8];
	if (c > 8) return false;
	if (c > 0) {
		if (tid < c * c) {
			const int a = tid / c, b2 = tid - a * c;
			const int ia = idx[a], ib = idx[b2];
			K[a * 8 + b2] = Minv[ia * ld + ib] + ((a == b2) ? 1.0 / (thr - Mm[ia * ld + ia]) : 0.0);
		}
		SYNC();
		if (tid == 0) {  // in-place Gauss-Jordan inverse of the SPD c x c matrix (no pivoting needed)
			for (int k = 0; k < c; k++) {
				const double rd = 1.0 / K[k * 8 + k];
				for (int j = 0; j < c; j++) K[k * 8 + j] = (j == k) ? rd : K[k * 8 + j] * rd;
				for (int i = 0; i < c; i++) {
					if (i == k) continue;
					const double f = K[i * 8 + k];
					for (int j = 0; j < c; j++) K[i * 8 + j] = (j == k) ? -f * rd : fma(-f, K[k * 8 + j], K[i * 8 + j]);
				}
			}
		}
		SYNC();
		for (int e = tid; e < n * c; e += T) {  // Z = Y K^-1
			const int i = e / c, a = e - i * c;
			double sacc = 0.0;
			for (int b2 = 0; b2 < c; b2++) sacc = fma(Minv[i * ld + idx[b2]], 0.5 * (K[b2 * 8 + a] + K[a * 8 + b2]), sacc);
			Z[i * ld + a] = sacc;
		}
		SYNC();
	}
	for (int e = tid; e < n * n; e += T) {
		const int i = e / n, j = e - i * n;
		const int lo = i < j ? i : j, hi = i < j ? j : i;  // the same expression for (i, j) and (j, i): bitwise symmetric like the sweep inverse
		double v = Minv[lo * ld + hi];
		for (int a = 0; a < c; a++) v = fma(-Z[lo * ld + a], Minv[hi * ld + idx[a]], v);
		MBinv[i * ld + j] = v;
	}
	SYNC();
	return true;
}

// Symmetric eigen-decomposition by parallel-ordered (round-robin) Jacobi: A -> diag (eigenvalues), V (optional) <- eigenvectors.
// cs: 4*(NMAX/2+1) doubles scratch, flag: one int in LDS.  3 barriers per round, n-1 (n even) or n rounds per sweep.
template <int T>
__device__ void jacobi_eig(double* A, double* V, int n, int ld, double* cs, int* flag, int tid) {
	if (V) {
		for (int e = tid; e < n * n; e += T) {
			int i = e / n, j = e - i * n;
			V[i * ld + j] = (i == j) ? 1.0 : 0.0;
		}
	}
	double tr = 0.0;
	for (int i = 0; i < n; i++) tr += fabs(A[i * ld + i]);
	const double thresh = 1e-15 * (tr / n);  // off-diagonals below this are zero for every downstream tolerance
	if (n < 2) {
		SYNC();
		return;
	}
	const int npad = (n + 1) & ~1, half = npad >> 1, rounds = npad - 1;
	for (int sweep = 0; sweep < 14; sweep++) {
		if (tid == 0) *flag = 0;
		SYNC();
		for (int r = 0; r < rounds; r++) {
			if (tid < half) {
				int p, q;
				if (tid == 0) {
					p = npad - 1;
					q = r;
				} else {
					p = (r + tid) % rounds;
					q = (r - tid + rounds) % rounds;
				}
				if (p > q) {
					int t_ = p;
					p = q;
					q = t_;
				}
				double c = 1.0, s = 0.0;
				if (q < n) {
					double app = A[p * ld + p], aqq = A[q * ld + q], apq = A[p * ld + q];
					if (fabs(apq) > thresh) {
						double theta = (aqq - app) / (2.0 * apq);
						double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
						c = 1.0 / sqrt(t * t + 1.0);
						s = t * c;
						*flag = 1;
					}
				} else {
					q = -1;
				}
				cs[4 * tid + 0] = c;
				cs[4 * tid + 1] = s;
				cs[4 * tid + 2] = (double)p;
				cs[4 * tid + 3] = (double)q;
			}
			SYNC();
			// column rotation A <- A J, V <- V J
			for (int e = tid; e < n * half; e += T) {
				int i = e / half, pi = e - i * half;
				int q = (int)cs[4 * pi + 3];
				if (q < 0) continue;
				int p = (int)cs[4 * pi + 2];
				double c = cs[4 * pi], s = cs[4 * pi + 1];
				double ap = A[i * ld + p], aq = A[i * ld + q];
				A[i * ld + p] = c * ap - s * aq;
				A[i * ld + q] = s * ap + c * aq;
				if (V) {
					double vp = V[i * ld + p], vq = V[i * ld + q];
					V[i * ld + p] = c * vp - s * vq;
					V[i * ld + q] = s * vp + c * vq;
				}
			}
			SYNC();
			// row rotation A <- J^T A
			for (int e = tid; e < n * half; e += T) {
				int pi = e / n, j = e - pi * n;
				int q = (int)cs[4 * pi + 3];
				if (q < 0) continue;
				int p = (int)cs[4 * pi + 2];
				double c = cs[4 * pi], s = cs[4 * pi + 1];
				double ap = A[p * ld + j], aq = A[q * ld + j];
				A[p * ld + j] = c * ap - s * aq;
				A[q * ld + j] = s * ap + c * aq;
			}
			SYNC();
		}
		if (*flag == 0) break;
		SYNC();
	}
}

__device__ __forceinline__ void mat3_mul_vec(const double* R, const double* v, double* o) {
	o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
	o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
	o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}


}  // namespace saip
