// micro-benchmark: v_fma_f64 issue rate, dependent latency and the clock the chip sustains under dense fp64 VALU load.
// Build: hipcc -O3 --offload-arch=gfx950 fp64bench.hip -o fp64bench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int CHAINS>
__global__ __launch_bounds__(512) void fma_kernel(double *out, long iters, long long *cycles)
{
	double a[CHAINS];
	const double c0 = 0.999 + threadIdx.x * 1e-9, c1 = 1e-3;
#pragma unroll
	for (int i = 0; i < CHAINS; ++i) a[i] = threadIdx.x + i;
	const long long t0 = __builtin_readcyclecounter();
	for (long it = 0; it < iters; ++it) {
#pragma unroll
		for (int r = 0; r < 16; ++r)
#pragma unroll
			for (int i = 0; i < CHAINS; ++i) a[i] = fma(a[i], c0, c1);
	}
	const long long t1 = __builtin_readcyclecounter();
	double s = 0;
#pragma unroll
	for (int i = 0; i < CHAINS; ++i) s += a[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int CHAINS> int run(int blocks, int threads, long iters, double *out, long long *dcyc)
{
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	fma_kernel<CHAINS><<<blocks, threads>>>(out, 10, dcyc);
	CHECK(hipEventRecord(e0));
	fma_kernel<CHAINS><<<blocks, threads>>>(out, iters, dcyc);
	CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
	float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
	long long cyc; CHECK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
	const double n_instr = (double) iters * 16 * CHAINS;     // per wave
	const int waves_per_simd = (blocks >= 256 ? (blocks / 256) : 1) * threads / 64 / 4;
	printf("chains=%d blocks=%d threads=%d (%d waves/SIMD): %.3f ms, counter %lld ticks -> %.3f GHz-equivalent if tick=shader clk; %.2f ticks/instr/wave; %.2f wall-ns per instr per SIMD; chip %.1f TFLOP/s\n",
	       CHAINS, blocks, threads, waves_per_simd, ms, cyc, cyc / (ms * 1e6), cyc / n_instr, ms * 1e6 / (n_instr * waves_per_simd),
	       2.0 * n_instr * 64 * (blocks * threads / 64) / (ms * 1e-3) / 1e12);
	return 0;
}

int main()
{
	double *out; long long *dcyc;
	CHECK(hipMalloc(&out, 8 * 1024 * 1024)); CHECK(hipMalloc(&dcyc, 8));
	run<1>(256, 256, 20000, out, dcyc);
	run<2>(256, 256, 20000, out, dcyc);
	run<4>(256, 256, 10000, out, dcyc);
	run<8>(256, 256, 10000, out, dcyc);
	run<1>(256, 512, 20000, out, dcyc);
	run<2>(256, 512, 20000, out, dcyc);
	run<4>(256, 512, 10000, out, dcyc);
	run<8>(256, 512, 10000, out, dcyc);
	run<1>(512, 512, 20000, out, dcyc);
	run<4>(512, 512, 10000, out, dcyc);
	return 0;
}
