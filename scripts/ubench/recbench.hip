// micro-benchmark: issue efficiency of the cascade's per-section recurrence loop (kernels_cascade.hip run_op_fast) at 1..4 waves/SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 recbench.hip -o recbench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int L = 16;

template <int MODE>
__global__ __launch_bounds__(256) void rec_kernel(double *out, const double *__restrict__ cf, int n_sec, long iters)
{
	double v[L];
#pragma unroll
	for (int i = 0; i < L; ++i) v[i] = 1e-3 * (threadIdx.x + i);
	double fx0 = 1e-5 * threadIdx.x, fx1 = 2e-5 * threadIdx.x, fnc3 = 0.5, fnc4 = -0.25;
	for (long it = 0; it < iters; ++it) {
		for (int j = 0; j < n_sec; ++j) {
			const double c0 = cf[8 * j + 0], c1 = cf[8 * j + 1], c2 = cf[8 * j + 2], nc3 = cf[8 * j + 3], nc4 = cf[8 * j + 4];
			double m0 = 0.0, m1 = 0.0;
			double x0 = fx0, x1 = fx1;
#pragma unroll
			for (int i = 0; i < L; ++i) {
				double s = v[i];
				if (MODE == 0) {
					s = v[i] + x0;
					const double t = fnc4 * x0;
					x0 = fma(fnc3, x0, x1);
					x1 = t;
				}
				const double r = fma(c0, s, m0);
				m0 = fma(nc3, r, fma(c1, s, m1));
				m1 = fma(nc4, r, c2 * s);
				v[i] = r;
			}
			fx0 = m0 * 1e-3; fx1 = m1 * 1e-3; fnc3 = nc3; fnc4 = nc4;
		}
	}
	double s = 0;
#pragma unroll
	for (int i = 0; i < L; ++i) s += v[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s + fx0 + fx1;
}

template <int MODE> int run(int blocks, int threads, long iters, double *out, const double *cf)
{
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	rec_kernel<MODE><<<blocks, threads>>>(out, cf, 10, 2);
	CHECK(hipEventRecord(e0));
	rec_kernel<MODE><<<blocks, threads>>>(out, cf, 10, iters);
	CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
	float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
	const int waves_per_simd = blocks / 256 * threads / 64 / 4 > 0 ? blocks / 256 * threads / 64 / 4 : 1;
	const double sec_tiles = (double) iters * 10;            // per wave
	const int ops = MODE == 0 ? 8 : 5;
	printf("mode=%d (%d f64 ops/sample) blocks=%d threads=%d (%d waves/SIMD): %.3f ms; %.1f ns per section-tile per SIMD-slot = %.0f cycles @2.4GHz per wave-section (ideal %d)\n",
	       MODE, ops, blocks, threads, waves_per_simd, ms, ms * 1e6 / sec_tiles / waves_per_simd, ms * 1e6 / sec_tiles / waves_per_simd * 2.4, ops * L * 4);
	return 0;
}

int main()
{
	double *out, *cf;
	CHECK(hipMalloc(&out, 8 * 1024 * 1024)); CHECK(hipMalloc(&cf, 8 * 80));
	double h[80]; for (int i = 0; i < 80; ++i) h[i] = 0.1 + 0.01 * i;
	CHECK(hipMemcpy(cf, h, sizeof(h), hipMemcpyHostToDevice));
	for (int w = 1; w <= 4; ++w) { run<0>(256 * w, 256, 2000, out, cf); run<1>(256 * w, 256, 2000, out, cf); }
	return 0;
}
