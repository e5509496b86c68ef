// micro-benchmark: one cascade SECTION (recurrence + cross-lane carry) in isolation, rows style (L frames per lane, 16-lane
// rows) at L = 16 / 32, against the bare recurrence.  Build: hipcc -O3 --offload-arch=gfx950 secbench.hip -o secbench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
	const long long b = __double_as_longlong(v);
	const int lo = __builtin_amdgcn_update_dpp(0, (int) (b & 0xffffffffLL), CTRL, 0xf, 0xf, true);
	const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), CTRL, 0xf, 0xf, true);
	return __longlong_as_double(((long long) hi << 32) | (unsigned int) lo);
}
__device__ __forceinline__ void row_scan(double &m0, double &m1, const double (&Pw)[16])
{
	double t0, t1;
	t0 = dpp_f64<0x111>(m0); t1 = dpp_f64<0x111>(m1);
	m0 += Pw[0] * t0 + Pw[1] * t1; m1 += Pw[2] * t0 + Pw[3] * t1;
	t0 = dpp_f64<0x112>(m0); t1 = dpp_f64<0x112>(m1);
	m0 += Pw[4] * t0 + Pw[5] * t1; m1 += Pw[6] * t0 + Pw[7] * t1;
	t0 = dpp_f64<0x114>(m0); t1 = dpp_f64<0x114>(m1);
	m0 += Pw[8] * t0 + Pw[9] * t1; m1 += Pw[10] * t0 + Pw[11] * t1;
	t0 = dpp_f64<0x118>(m0); t1 = dpp_f64<0x118>(m1);
	m0 += Pw[12] * t0 + Pw[13] * t1; m1 += Pw[14] * t0 + Pw[15] * t1;
}

// MODE 0: recurrence only; 1: + carry (inject, row scan, shift)
template <int L, int MODE, int WPE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void sec_kernel(double *out, const double *__restrict__ cf, int n_sec, long iters)
{
	__shared__ double st[2 * 8 * 16];
	double v[L];
#pragma unroll
	for (int i = 0; i < L; ++i) v[i] = 1e-3 * (threadIdx.x + i);
	const int q = threadIdx.x & 15;
	double *st_row = st + (threadIdx.x >> 4) * 32;
	if (threadIdx.x < 256) st[threadIdx.x % 256] = 0.0;
	__syncthreads();
	double fx0 = 0, fx1 = 0, fnc3 = 0, fnc4 = 0;
	for (long it = 0; it < iters; ++it) {
		for (int j = 0; j < n_sec; ++j) {
			const double *__restrict__ od = cf + 32 * j;
			double Pw[16];
#pragma unroll
			for (int i = 0; i < 16; ++i) Pw[i] = od[8 + i];
			const double2 xin = *reinterpret_cast<const double2 *>(st_row + 2 * j);
			__builtin_amdgcn_sched_barrier(0);
			const double c0 = od[2], c1 = od[3], c2 = od[4], nc3 = -od[5], nc4 = -od[6];
			double m0 = 0.0, m1 = 0.0;
			double x0 = fx0, x1 = fx1;
#pragma unroll
			for (int i = 0; i < L; ++i) {
				const double s = v[i] + x0;
				const double t = fnc4 * x0;
				x0 = fma(fnc3, x0, x1);
				x1 = t;
				const double r = fma(c0, s, m0);
				m0 = fma(nc3, r, fma(c1, s, m1));
				m1 = fma(nc4, r, c2 * s);
				v[i] = r;
			}
			__builtin_amdgcn_sched_barrier(0);
			if (MODE == 1) {
				const double e0 = fma(Pw[0], xin.x, fma(Pw[1], xin.y, m0)), e1 = fma(Pw[2], xin.x, fma(Pw[3], xin.y, m1));
				if (q == 0) { m0 = e0; m1 = e1; }
				row_scan(m0, m1, Pw);
				double y0 = dpp_f64<0x111>(m0), y1 = dpp_f64<0x111>(m1);
				if (q == 0) { y0 = xin.x; y1 = xin.y; }
				fx0 = y0; fx1 = y1;
				if (q == 15) *reinterpret_cast<double2 *>(st_row + 2 * j) = make_double2(m0 * 1e-3, m1 * 1e-3);
			}
			else { fx0 = m0 * 1e-3; fx1 = m1 * 1e-3; }
			fnc3 = nc3; fnc4 = nc4;
		}
	}
	double s = 0;
#pragma unroll
	for (int i = 0; i < L; ++i) s += v[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s + fx0 + fx1;
}

template <int L, int MODE, int WPE> int run(long iters, double *out, const double *cf)
{
	const int blocks = 256 * 2 * WPE;                        // 128 threads = 2 waves; WPE waves per SIMD = 4 WPE per CU
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	sec_kernel<L, MODE, WPE><<<blocks, 128>>>(out, cf, 10, 2);
	CHECK(hipEventRecord(e0));
	sec_kernel<L, MODE, WPE><<<blocks, 128>>>(out, cf, 10, iters);
	CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
	float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
	const double wave_sections_per_simd = (double) iters * 10 * WPE;
	printf("L=%d mode=%d waves/SIMD=%d: %.3f ms; %.1f ns per wave-section of SIMD time; %.3f ns per sample-lane (64 L samples per wave-section)\n",
	       L, MODE, WPE, ms, ms * 1e6 / wave_sections_per_simd, ms * 1e6 / wave_sections_per_simd / L);
	return 0;
}

int main()
{
	double *out, *cf;
	CHECK(hipMalloc(&out, 8 * 1024 * 1024)); CHECK(hipMalloc(&cf, 8 * 320));
	double h[320]; for (int i = 0; i < 320; ++i) h[i] = 0.05 + 0.001 * i;
	CHECK(hipMemcpy(cf, h, sizeof(h), hipMemcpyHostToDevice));
	run<16, 0, 2>(2000, out, cf); run<16, 1, 2>(2000, out, cf);
	run<32, 0, 2>(1000, out, cf); run<32, 1, 2>(1000, out, cf);
	run<16, 1, 3>(2000, out, cf); run<32, 1, 3>(1000, out, cf);
	run<16, 1, 4>(2000, out, cf); run<16, 1, 1>(2000, out, cf); run<32, 1, 1>(1000, out, cf);
	return 0;
}
