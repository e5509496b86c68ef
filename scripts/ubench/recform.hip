// micro-benchmark: the biquad recurrence in two algebraically equal forms, 2 waves per SIMD, 32 samples per lane and step
//   A (TDF-II as the reference writes it):  r = c0 x + m0;  m0' = nc3 r + (c1 x + m1);  m1' = nc4 r + c2 x      chain m0 -> r -> m0': 2 FMAs
//   B (state-space):                        r = c0 x + m0;  m0' = nc3 m0 + (k1 x + m1); m1' = nc4 m0 + k2 x     chain m0 -> m0': 1 FMA
// Build: hipcc -O3 --offload-arch=gfx950 recform.hip -o recform
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int FORM, int FIX>
__global__ __launch_bounds__(512) void k(double *out, long steps)
{
	double v[32];
#pragma unroll
	for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3 + i;
	const double c0 = 0.999, c1 = 1e-3, nc3 = -0.5, nc4 = 0.25, c2 = 0.1, k1 = c1 + nc3 * c0, k2 = c2 + nc4 * c0;
	double m0 = 0.0, m1 = 0.0, x0 = 1e-3, x1 = 2e-3;
	for (long s = 0; s < steps; ++s) {
#pragma unroll
		for (int i = 0; i < 32; ++i) {
			double x = v[i];
			if (FIX) { x = x + x0; const double t = nc4 * x0; x0 = fma(nc3, x0, x1); x1 = t; }
			const double r = fma(c0, x, m0);
			if (FORM == 0) { m0 = fma(nc3, r, fma(c1, x, m1)); m1 = fma(nc4, r, c2 * x); }
			else { const double n0 = fma(nc3, m0, fma(k1, x, m1)); m1 = fma(nc4, m0, k2 * x); m0 = n0; }
			v[i] = r;
		}
	}
	double sum = m0 + m1 + x0 + x1;
#pragma unroll
	for (int i = 0; i < 32; ++i) sum += v[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int FORM, int FIX> int run(const char *name, double *out)
{
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	const long steps = 20000;
	k<FORM, FIX><<<256, 512>>>(out, 10);
	CHECK(hipEventRecord(e0));
	k<FORM, FIX><<<256, 512>>>(out, steps);
	CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
	float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
	printf("%-44s %8.3f ms  = %7.1f ns per step and SIMD (2 waves)\n", name, ms, ms * 1e6 / steps);
	return 0;
}

int main()
{
	double *out; CHECK(hipMalloc(&out, 8 * 1024 * 1024));
	run<0, 0>("TDF-II, 5 instr/sample", out);
	run<1, 0>("state-space, 5 instr/sample", out);
	run<0, 1>("TDF-II + pending fix, 8 instr/sample", out);
	run<1, 1>("state-space + pending fix, 8 instr/sample", out);
	return 0;
}
