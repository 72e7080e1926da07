// microbenchmark (diagnostic): what one wavefront alone on a SIMD sustains in FP64 on gfx950, by s_memtime inside the kernel.
//   hipcc --offload-arch=gfx950 -O3 -o fp64_issue tools/ubench/fp64_issue.hip && ./fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, unsigned long long* t, int active, double seed) {
	const int lane = threadIdx.x;
	double a0 = seed + lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	const double m = 1.0000001, c = 1e-9;
	unsigned long long t0 = 0, t1 = 0;
	if (lane < active) {
		__builtin_amdgcn_sched_barrier(0);
		asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
		for (int it = 0; it < 16; it++) {
			if (MODE == 0) {  // 8 independent fma chains
#pragma unroll
				for (int r = 0; r < REP; r++) {
					a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
					a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
				}
			} else if (MODE == 1) {  // one dependent chain
#pragma unroll
				for (int r = 0; r < REP * 8; r++) a0 = fma(a0, m, c);
			} else if (MODE == 2) {  // two dependent chains
#pragma unroll
				for (int r = 0; r < REP * 4; r++) { a0 = fma(a0, m, c); a1 = fma(a1, m, c); }
			} else if (MODE == 3) {  // rsq chain
#pragma unroll
				for (int r = 0; r < REP * 8; r++) a0 = __builtin_amdgcn_rsq(a0 + 2.0);
			} else if (MODE == 4) {  // independent rcp
#pragma unroll
				for (int r = 0; r < REP; r++) {
					a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3);
					a4 = __builtin_amdgcn_rcp(a4); a5 = __builtin_amdgcn_rcp(a5); a6 = __builtin_amdgcn_rcp(a6); a7 = __builtin_amdgcn_rcp(a7);
				}
			} else if (MODE == 5) {  // DPP row_shr moves of a double (2 movs), dependent
#pragma unroll
				for (int r = 0; r < REP * 8; r++) {
					int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a0), 0x111, 0xF, 0xF, true);
					int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a0), 0x111, 0xF, 0xF, true);
					a0 = __hiloint2double(hi, lo);
				}
			} else if (MODE == 6) {  // 4 independent chains
#pragma unroll
				for (int r = 0; r < REP * 2; r++) { a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c); }
			} else if (MODE == 7) {  // f32 independent fma for comparison
				float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
#pragma unroll
				for (int r = 0; r < REP; r++) {
					f0 = fmaf(f0, 1.0001f, 1e-3f); f1 = fmaf(f1, 1.0001f, 1e-3f); f2 = fmaf(f2, 1.0001f, 1e-3f); f3 = fmaf(f3, 1.0001f, 1e-3f);
					f4 = fmaf(f4, 1.0001f, 1e-3f); f5 = fmaf(f5, 1.0001f, 1e-3f); f6 = fmaf(f6, 1.0001f, 1e-3f); f7 = fmaf(f7, 1.0001f, 1e-3f);
				}
				a0 = f0; a1 = f1; a2 = f2; a3 = f3; a4 = f4; a5 = f5; a6 = f6; a7 = f7;
			}
		}
		__builtin_amdgcn_sched_barrier(0);
		asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
		__builtin_amdgcn_sched_barrier(0);
	}
	out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
	if (lane == 0) t[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks, int threads_active, double* out, unsigned long long* t) {
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, t, threads_active, 1.0);
	hipDeviceSynchronize();
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, t, threads_active, 1.0);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	std::vector<unsigned long long> h(blocks);
	hipMemcpy(h.data(), t, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
	std::sort(h.begin(), h.end());
	const double n = 16.0 * REP * 8;
	printf("%-34s blocks %5d active %2d: median %7llu clk = %.2f clk/instr (kernel %.1f us)\n", name, blocks, threads_active, h[blocks / 2], h[blocks / 2] / n, ms * 1e3);
}
#include <algorithm>
int main() {
	double* out; unsigned long long* t;
	hipMalloc(&out, 8192 * 64 * 8); hipMalloc(&t, 8192 * 8);
	for (int blocks : {256, 1024, 2048, 4096}) {
		for (int act : {64, 32, 16}) {
			run<0>("fma f64 x8 independent", blocks, act, out, t);
		}
	}
	for (int act : {64, 32}) {
		run<1>("fma f64 dependent chain", 1024, act, out, t);
		run<2>("fma f64 2 chains", 1024, act, out, t);
		run<6>("fma f64 4 chains", 1024, act, out, t);
		run<3>("rsq f64 dependent (+add)", 1024, act, out, t);
		run<4>("rcp f64 x8 independent", 1024, act, out, t);
		run<5>("dpp mov pair dependent", 1024, act, out, t);
		run<7>("fma f32 x8 independent", 1024, act, out, t);
	}
	run<1>("fma f64 dependent chain", 2048, 64, out, t);
	run<1>("fma f64 dependent chain", 4096, 64, out, t);
	return 0;
}
