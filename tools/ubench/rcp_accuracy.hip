// How good are v_rcp_f64 / v_rsq_f64 on gfx950, and how many Newton steps do oct_rcp / oct_rsqrt need?  (diagnostic)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/rcp_accuracy.hip -o tools/ubench/rcp_accuracy && tools/ubench/rcp_accuracy
// Prints the largest relative error of the raw estimate and of the estimate after 1 and 2 Newton steps over 4 M arguments spread log-uniformly
// over [1e-8, 1e8], against long-double host values.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k(const double* x, double* out, int n) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double v = x[i];
	double y = __builtin_amdgcn_rcp(v);
	out[i] = y;
	y = fma(fma(-v, y, 1.0), y, y);
	out[n + i] = y;
	y = fma(fma(-v, y, 1.0), y, y);
	out[2 * n + i] = y;
	double s = __builtin_amdgcn_rsq(v);
	const double h = 0.5 * v;
	out[3 * n + i] = s;
	s = fma(s, fma(-h * s, s, 0.5), s);
	out[4 * n + i] = s;
	s = fma(s, fma(-h * s, s, 0.5), s);
	out[5 * n + i] = s;
}

int main() {
	const int n = 1 << 22;
	std::vector<double> x(n), o(6 * (size_t)n);
	srand(7);
	for (int i = 0; i < n; i++) x[i] = pow(10.0, -8.0 + 16.0 * (rand() / (double)RAND_MAX)) * (1.0 + rand() / (double)RAND_MAX);
	double *dx, *dout;
	hipMalloc(&dx, n * 8);
	hipMalloc(&dout, 6 * (size_t)n * 8);
	hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
	k<<<n / 256, 256>>>(dx, dout, n);
	hipMemcpy(o.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
	const char* names[6] = {"rcp estimate", "rcp + 1 Newton", "rcp + 2 Newton", "rsq estimate", "rsq + 1 Newton", "rsq + 2 Newton"};
	for (int c = 0; c < 6; c++) {
		long double worst = 0;
		for (int i = 0; i < n; i++) {
			const long double ref = c < 3 ? 1.0L / (long double)x[i] : 1.0L / sqrtl((long double)x[i]);
			const long double e = fabsl(((long double)o[(size_t)c * n + i] - ref) / ref);
			if (e > worst) worst = e;
		}
		printf("%-16s max relative error %.3Le  (%.2Lf bits)\n", names[c], worst, -log2l(worst));
	}
	return 0;
}
