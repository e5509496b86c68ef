// micro-benchmark: streaming read / copy bandwidth as a function of working-set size (is the 256 MiB
// Infinity Cache faster than HBM for re-read data?).  Build: hipcc -O3 --offload-arch=gfx950 mallbench.hip -o mallbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void rd(const double2 *__restrict__ p, size_t n, double *sink)
{
	double2 a = make_double2(0, 0);
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
		double2 v = p[i]; a.x += v.x; a.y += v.y;
	}
	if (a.x == 1.2345) *sink = a.y;
}
__global__ __launch_bounds__(256) void cp(const double2 *__restrict__ p, double2 *__restrict__ q, size_t n)
{
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) q[i] = p[i];
}
__global__ __launch_bounds__(256) void rmw(double2 *__restrict__ p, size_t n)
{
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) { double2 v = p[i]; v.x += 1.0; p[i] = v; }
}
int main()
{
	const size_t maxb = (size_t) 4 << 30;
	double2 *a, *b; double *sink;
	hipMalloc(&a, maxb); hipMalloc(&b, maxb); hipMalloc(&sink, 8);
	hipMemset(a, 0, maxb); hipMemset(b, 0, maxb);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const size_t sizes_mb[] = { 16, 32, 64, 96, 128, 160, 192, 256, 384, 512, 1024, 4096 };
	for (size_t mb : sizes_mb) {
		const size_t bytes = mb << 20, n = bytes / 16;
		const int reps = (int) ((size_t) 16384 / mb) + 2;
		for (int mode = 0; mode < 3; ++mode) {
			for (int w = 0; w < 2; ++w) {
				if (w == 1) hipEventRecord(e0);
				for (int r = 0; r < reps; ++r) {
					if (mode == 0) rd<<<4096, 256>>>(a, n, sink);
					else if (mode == 1) cp<<<4096, 256>>>(a, b, n);
					else rmw<<<4096, 256>>>(a, n);
				}
			}
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			const double gb = (double) bytes * reps * (mode == 0 ? 1 : 2) / 1e9;
			printf("%5zu MiB %s : %8.1f GB/s (%d reps, %.3f ms each)\n", mb, mode == 0 ? "read " : mode == 1 ? "copy " : "rmw  ", gb / (ms * 1e-3), reps, ms / reps);
		}
	}
	return 0;
}
