// FETCH_SIZE / WRITE_SIZE calibration on gfx950: kernels that move a KNOWN number of bytes with 4, 8 and 16 bytes per lane.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out/f -o run --output-format csv -- tools/ubench/fetch_calib
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out/w -o run --output-format csv -- tools/ubench/fetch_calib
//   python3 tools/ubench/fetch_calib_report.py out        (prints counter KiB / true KiB per kernel)
//
// MI355X_MICROARCH.md:298 states the factor for 16 B/lane streaming reads (FETCH_SIZE = half the bytes).  The cycle kernels read their state as
// one double per lane (8 B/lane, 512 B per wavefront request), so the factor applied in profiles/pmc_traffic.json has to be measured for THAT width.
// Every kernel touches each byte of its range exactly once; the big range (1 GiB) is past every cache level, the small one (4 MiB, what one
// launch of the headline kernel reads) is first-touch after a 1 GiB eviction sweep.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename T>
__global__ void read_stream(const T* __restrict__ src, size_t count, double* sink) {
	T acc{};
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
		T v = src[i];
		if constexpr (sizeof(T) == 16) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
		else acc += v;
	}
	double s;
	if constexpr (sizeof(T) == 16) s = (double)acc.x + acc.y + acc.z + acc.w; else s = (double)acc;
	if (s == 12345.0) *sink = s;  // (all zeros in the buffers: never true, but not provably so) keeps the loads
}

template <typename T>
__global__ void write_stream(T* __restrict__ dst, size_t count) {
	T v{};
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
		dst[i] = v;
}

// the state layout of the cycle kernels: [rows][ld] doubles, instance index fastest; a wavefront reads 64 consecutive doubles of row r, then row r+1 ...
__global__ void read_rows(const double* __restrict__ src, int rows, int ld, double* sink) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	double acc = 0;
	if (i < ld)
		for (int r = 0; r < rows; ++r) acc += src[(size_t)r * ld + i];
	if (acc == 1.2345e301) *sink = acc;
}

int main() {
	const size_t big = (size_t)1 << 30, small = (size_t)4 << 20;
	char *a, *b;
	double* sink;
	CK(hipMalloc(&a, big));
	CK(hipMalloc(&b, big));
	CK(hipMalloc(&sink, 8));
	CK(hipMemset(a, 0, big));
	CK(hipMemset(b, 0, big));
	const int grid = 256 * 8, blk = 256;
	for (int rep = 0; rep < 3; ++rep) {
		read_stream<float><<<grid, blk>>>((const float*)a, big / 4, sink);    // 1 GiB, 4 B/lane
		read_stream<double><<<grid, blk>>>((const double*)b, big / 8, sink);  // 1 GiB, 8 B/lane
		read_stream<float4><<<grid, blk>>>((const float4*)a, big / 16, sink); // 1 GiB, 16 B/lane
		write_stream<float><<<grid, blk>>>((float*)b, big / 4);
		write_stream<double><<<grid, blk>>>((double*)a, big / 8);
		write_stream<float4><<<grid, blk>>>((float4*)b, big / 16);
		// small first-touch ranges: evict with a 1 GiB sweep of the OTHER buffer first
		read_stream<float4><<<grid, blk>>>((const float4*)b, big / 16, sink);
		read_stream<double><<<64, blk>>>((const double*)a, small / 8, sink);  // 4 MiB, 8 B/lane
		read_stream<float4><<<grid, blk>>>((const float4*)b, big / 16, sink);
		read_rows<<<4096 / 64, 64>>>((const double*)a, 128, 4096, sink);       // 128 rows x 4096 doubles = 4 MiB, the kernels' access shape
		read_stream<float4><<<grid, blk>>>((const float4*)b, big / 16, sink);
		write_stream<double><<<64, blk>>>((double*)a, small / 8);              // 4 MiB written, 8 B/lane
	}
	CK(hipDeviceSynchronize());
	printf("done: big = %zu bytes, small = %zu bytes\n", big, small);
	return 0;
}
