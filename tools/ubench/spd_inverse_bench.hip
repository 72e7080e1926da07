// microbenchmark (diagnostic): cycles of the workgroup-cooperative SPD inverse of saip_wg_linalg.h in the launch shape of the general kernel
// (256 threads, one workgroup per CU), plus the cost of a bare barrier loop and of a barrier + dependent LDS round trip.
//   hipcc --offload-arch=gfx950 -O3 -I sai-primitives_amd/csrc -o tools/ubench/spd_inverse_bench tools/ubench/spd_inverse_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "saip_wg_linalg.h"
using namespace saip;

constexpr int LD = 33, MS = 32 * 33;
__global__ void __launch_bounds__(256) k(const double* Ain, double* Aout, unsigned long long* t, int n, int mode) {
	__shared__ double A[MS], B[MS], W1[MS], W2[MS], pad[12 * MS];  // ~135 KB like the general kernel: one workgroup per CU
	const int tid = threadIdx.x;
	for (int e = tid; e < n * n; e += 256) A[(e / n) * LD + e % n] = Ain[e];
	pad[tid] = 0;
	__syncthreads();
	unsigned long long t0, t1;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
	if (mode == 0) {
		spd_inverse<256>(A, B, W1, W2, nullptr, n, LD, tid);
	} else if (mode == 1) {
		for (int i = 0; i < n; i++) __syncthreads();
	} else if (mode == 2) {
		double x = A[tid % 30];
		for (int i = 0; i < n; i++) {
			W1[tid] = x;
			__syncthreads();
			x += W1[(tid + 1 + i) & 255];
			__syncthreads();
		}
		B[tid % 30] = x;
	}
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
	for (int e = tid; e < n * n; e += 256) Aout[blockIdx.x * 1024 + e] = B[(e / n) * LD + e % n];
	if (tid == 0) t[blockIdx.x] = t1 - t0;
}
int main() {
	const int n = 30;
	std::vector<double> h(n * n), hi(1024);
	for (int i = 0; i < n; i++)
		for (int j = 0; j < n; j++) h[i * n + j] = (i == j ? 4.0 + i * 0.1 : 0.0) + 1.0 / (1.0 + abs(i - j));
	double *Ain, *Aout; unsigned long long* t;
	hipMalloc(&Ain, n * n * 8); hipMalloc(&Aout, 256 * 1024 * 8); hipMalloc(&t, 256 * 8);
	hipMemcpy(Ain, h.data(), n * n * 8, hipMemcpyHostToDevice);
	for (int mode = 0; mode < 3; mode++)
		for (int rep = 0; rep < 2; rep++) {
			hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, Ain, Aout, t, n, mode);
			hipDeviceSynchronize();
			unsigned long long ht[256];
			hipMemcpy(ht, t, sizeof(ht), hipMemcpyDeviceToHost);
			printf("mode %d: %llu clk total, %.0f per step\n", mode, ht[128], ht[128] / (double)n);
		}
	hipMemcpy(hi.data(), Aout, 1024 * 8, hipMemcpyDeviceToHost);
	double err = 0;  // A * Ainv = I ?
	for (int i = 0; i < n; i++)
		for (int j = 0; j < n; j++) {
			double s = 0;
			for (int l = 0; l < n; l++) s += h[i * n + l] * hi[l * n + j];
			err = fmax(err, fabs(s - (i == j)));
		}
	printf("|A Ainv - I| = %.2e (mode 2 ran last: ignore unless mode order changed)\n", err);
	return 0;
}
