// micro-benchmark: does v_mfma_f64_16x16x4 run beside fp64 VALU work of the same wave?  (the cascade's zero-input correction as
// a rank-2 update on the matrix pipe: docs/history.md section 8)   Build: hipcc -O3 --offload-arch=gfx950 mfmabench.hip -o mfmabench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

// MODE 1: VALU only (NV dependent-chain FMAs per "sample", 32 samples per step), MODE 2: MFMA only (NM per step), MODE 3: both
template <int MODE, int NM>
__global__ __launch_bounds__(512) void k(double *out, long steps)
{
	double v[32];
#pragma unroll
	for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3 + i;
	v4d acc[8];
#pragma unroll
	for (int g = 0; g < 8; ++g) acc[g] = (v4d) { 0.0, 0.0, 0.0, 0.0 };
	double a = 1e-3 * threadIdx.x, b = 0.5 + 1e-4 * threadIdx.x;
	const double c0 = 0.999, c1 = 1e-3, nc3 = -0.5, nc4 = 0.25, c2 = 0.1;
	double m0 = 0.0, m1 = 0.0;
	for (long s = 0; s < steps; ++s) {
		if (MODE & 2) {
#pragma unroll
			for (int q = 0; q < NM; ++q) acc[q & 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q & 7], 0, 0, 0);
		}
		if (MODE & 1) {
#pragma unroll
			for (int i = 0; i < 32; ++i) {
				const double x = v[i];
				const double r = fma(c0, x, m0);
				m0 = fma(nc3, r, fma(c1, x, m1));
				m1 = fma(nc4, r, c2 * x);
				v[i] = r;
			}
		}
	}
	double sum = m0 + m1;
#pragma unroll
	for (int i = 0; i < 32; ++i) sum += v[i];
#pragma unroll
	for (int g = 0; g < 8; ++g) sum += acc[g].x + acc[g].y + acc[g].z + acc[g].w;
	out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int MODE, int NM> int run(const char *name, double *out)
{
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	const long steps = 20000;
	k<MODE, NM><<<256, 512>>>(out, 10);
	CHECK(hipEventRecord(e0));
	k<MODE, NM><<<256, 512>>>(out, steps);
	CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
	float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
	printf("%-28s %8.3f ms  = %7.1f ns per step per wave-pair (2 waves/SIMD)\n", name, ms, ms * 1e6 / steps);
	return 0;
}

int main()
{
	double *out; CHECK(hipMalloc(&out, 8 * 1024 * 1024));
	run<1, 0>("VALU only (160 FMA/step)", out);
	run<2, 8>("MFMA only, 8 per step", out);
	run<2, 12>("MFMA only, 12 per step", out);
	run<2, 16>("MFMA only, 16 per step", out);
	run<3, 8>("VALU + 8 MFMA", out);
	run<3, 12>("VALU + 12 MFMA", out);
	run<3, 16>("VALU + 16 MFMA", out);
	return 0;
}
