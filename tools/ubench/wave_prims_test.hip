// GPU check of csrc/saip_wave_prims.h (one wavefront): stored form, wv_mm, wv_sweep, wv_matvec, the row / wave reductions, against
// host arithmetic.  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -I sai-primitives_amd/csrc tools/ubench/wave_prims_test.hip -o gpurun_out/wave_prims_test && gpurun_out/wave_prims_test
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "saip_wave_prims.h"
using namespace saip;

// in: A (32 x 32 row-major, SPD n x n padded with the identity), Bm (32 x 32), x (32);  out: [0] -inv(A) swept, [1] A^T B, [2] (A^T x | hsum | rowsum | argmax)
__global__ void __launch_bounds__(64) k_test(const double* A, const double* Bm, const double* x, double* out, int n) {
	const int lane = threadIdx.x, lo = lane & 15, hi = lane >> 4;
	double a[8][2], b[8][2], c[8][2];
	for (int ks = 0; ks < 8; ks++)
		for (int t = 0; t < 2; t++) {
			a[ks][t] = A[(4 * ks + hi) * 32 + 16 * t + lo];
			b[ks][t] = Bm[(4 * ks + hi) * 32 + 16 * t + lo];
		}
	wv_mm<false, false>(a, b, c);
	for (int ks = 0; ks < 8; ks++)
		for (int t = 0; t < 2; t++) out[1024 + (4 * ks + hi) * 32 + 16 * t + lo] = c[ks][t];
	double xr[8], y[2];
	wv_load_rows(x, hi, xr);
	wv_matvec(a, xr, y);
	if (hi == 0) {
		out[2048 + lo] = y[0];
		out[2048 + 16 + lo] = y[1];
	}
	out[2048 + 64 + lane] = wv_hsum(x[lane & 31] * (1 + hi));
	out[2048 + 128 + lane] = wv_rowsum(x[lane & 31] + hi);
	double best = x[lo] * (hi + 1);
	int jb = lo;
	wv_row_argmax(best, jb);
	out[2048 + 192 + lane] = best;
	out[2048 + 256 + lane] = jb;
	const double mp = wv_sweep<2>(a, (n + 3) / 4, lo, hi);
	for (int ks = 0; ks < 8; ks++)
		for (int t = 0; t < 2; t++) out[(4 * ks + hi) * 32 + 16 * t + lo] = a[ks][t];
	if (lane == 0) out[2048 + 320] = mp;
	// one-tile sweep of the leading 16 x 16 block of B B^T + I (k = 6 padded to 8)
	double g[4][1], g2[4][1];
	for (int ks = 0; ks < 4; ks++) g[ks][0] = Bm[1024 + (4 * ks + hi) * 16 + lo];
	const double mp1 = wv_sweep<1>(g, 2, lo, hi);
	for (int ks = 0; ks < 4; ks++) out[3072 + (4 * ks + hi) * 16 + lo] = g[ks][0];
	if (lane == 0) out[2048 + 321] = mp1;
	(void)g2;
}

int main() {
	const int n = 30;
	std::vector<double> A(1024, 0.0), Bm(1024 + 256, 0.0), x(32), out(4096, 0.0);
	srand(3);
	auto rnd = []() { return (double)rand() / RAND_MAX - 0.5; };
	std::vector<double> R(1024);
	for (auto& v : R) v = rnd();
	for (int i = 0; i < 32; i++)
		for (int j = 0; j < 32; j++) {
			double s = 0;
			if (i < n && j < n) {
				for (int k = 0; k < 32; k++) s += R[i * 32 + k] * R[j * 32 + k];
				if (i == j) s += 1.0;
			} else if (i == j) s = 1.0;
			A[i * 32 + j] = s;
		}
	for (int i = 0; i < 1024; i++) Bm[i] = rnd();
	// 16 x 16 tile: 6 x 6 SPD, identity on 6, 7, zero beyond
	for (int i = 0; i < 16; i++)
		for (int j = 0; j < 16; j++) {
			double s = 0;
			if (i < 6 && j < 6) {
				for (int k = 0; k < 32; k++) s += R[i * 32 + k] * R[j * 32 + k];
			} else if (i == j && i < 8) s = 1.0;
			Bm[1024 + i * 16 + j] = s;
		}
	for (auto& v : x) v = rnd();
	double *dA, *dB, *dx, *dout;
	hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, Bm.size() * 8); hipMalloc(&dx, 32 * 8); hipMalloc(&dout, out.size() * 8);
	hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
	hipMemcpy(dB, Bm.data(), Bm.size() * 8, hipMemcpyHostToDevice);
	hipMemcpy(dx, x.data(), 32 * 8, hipMemcpyHostToDevice);
	hipMemset(dout, 0, out.size() * 8);
	hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, dA, dB, dx, dout, n);
	if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
	hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost);
	int bad = 0;
	// A^T B
	double e = 0;
	for (int i = 0; i < 32; i++)
		for (int j = 0; j < 32; j++) {
			double s = 0;
			for (int k = 0; k < 32; k++) s += A[k * 32 + i] * Bm[k * 32 + j];
			e = fmax(e, fabs(s - out[1024 + i * 32 + j]));
		}
	printf("wv_mm       max error %.3e\n", e); bad += e > 1e-10;
	e = 0;
	for (int j = 0; j < 32; j++) {
		double s = 0;
		for (int k = 0; k < 32; k++) s += A[k * 32 + j] * x[k];
		e = fmax(e, fabs(s - out[2048 + j]));
	}
	printf("wv_matvec   max error %.3e\n", e); bad += e > 1e-10;
	e = 0;
	for (int l = 0; l < 64; l++) {
		double s = 0;
		for (int h = 0; h < 4; h++) s += x[(l & 15) + 16 * (h & 1)] * (1 + h);
		e = fmax(e, fabs(s - out[2048 + 64 + l]));
	}
	printf("wv_hsum     max error %.3e\n", e); bad += e > 1e-12;
	e = 0;
	for (int l = 0; l < 64; l++) {
		double s = 0;
		for (int i = 0; i < 16; i++) s += x[i + 16 * ((l >> 4) & 1)] + (l >> 4);
		e = fmax(e, fabs(s - out[2048 + 128 + l]));
	}
	printf("wv_rowsum   max error %.3e\n", e); bad += e > 1e-12;
	e = 0;
	for (int l = 0; l < 64; l++) {
		int h = l >> 4, jb = 0;
		double best = -1e300;
		for (int i = 0; i < 16; i++)
			if (x[i] * (h + 1) > best) best = x[i] * (h + 1), jb = i;
		e = fmax(e, fabs(best - out[2048 + 192 + l]) + fabs(jb - out[2048 + 256 + l]));
	}
	printf("wv_argmax   max error %.3e\n", e); bad += e > 0;
	// -inv(A): A * (-out) = I on the leading n x n
	e = 0;
	for (int i = 0; i < 32; i++)
		for (int j = 0; j < 32; j++) {
			double s = 0;
			for (int k = 0; k < 32; k++) s += A[i * 32 + k] * (-out[k * 32 + j]);
			if (i < 4 * ((n + 3) / 4) && j < 4 * ((n + 3) / 4)) e = fmax(e, fabs(s - (i == j)));
		}
	printf("wv_sweep<2> |A inv(A) - I| %.3e  min pivot %.3g\n", e, out[2048 + 320]); bad += e > 1e-9;
	e = 0;
	for (int i = 0; i < 8; i++)
		for (int j = 0; j < 8; j++) {
			double s = 0;
			for (int k = 0; k < 8; k++) s += Bm[1024 + i * 16 + k] * (-out[3072 + k * 16 + j]);
			e = fmax(e, fabs(s - (i == j)));
		}
	printf("wv_sweep<1> |G inv(G) - I| %.3e  min pivot %.3g\n", e, out[2048 + 321]); bad += e > 1e-9;
	printf(bad ? "FAILED\n" : "all primitives OK\n");
	return bad;
}
