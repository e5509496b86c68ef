// rocfft_baseline.hip -- the library route the north star names, timed on the headline convolution problem:
// 1024 channel pairs x one 2^18-point complex fp64 transform each (the window of an overlap-save block of a 65536-tap
// filter), forward FFT (hipFFT / rocFFT Z2Z, batched, in place) -> pointwise multiply by the filter spectrum -> inverse FFT.
// This is the FFT part of one bench step (what conv_col_fwd + conv_row + conv_col_inv do in three trips); the gather of the
// window from the rings and the scatter of the valid 3/4 to the slab would come on top (two more passes).
// Build: hipcc -O3 --offload-arch=gfx950 rocfft_baseline.hip -o rocfft_baseline -lhipfft
// Run:   ./rocfft_baseline [log2N = 18] [batch = 1024]
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void mul_spectrum(double2 *w, const double2 *__restrict__ h, size_t n, size_t total)
{
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t) gridDim.x * blockDim.x) {
		const double2 a = w[i], b = h[i % n];
		w[i] = make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
	}
}

int main(int argc, char **argv)
{
	const int log2n = argc > 1 ? atoi(argv[1]) : 18, batch = argc > 2 ? atoi(argv[2]) : 1024;
	const size_t n = (size_t) 1 << log2n, total = n * batch;
	double2 *w, *h;
	if (hipMalloc(&w, total * 16) != hipSuccess || hipMalloc(&h, n * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipMemset(w, 0, total * 16);
	hipMemset(h, 0, n * 16);
	hipfftHandle plan;
	int nn[1] = { (int) n };
	if (hipfftPlanMany(&plan, 1, nn, nullptr, 1, (int) n, nullptr, 1, (int) n, HIPFFT_Z2Z, batch) != HIPFFT_SUCCESS) { printf("plan failed\n"); return 1; }
	size_t ws = 0;
	hipfftGetSize(plan, &ws);
	hipEvent_t e[4];
	for (auto &x : e) hipEventCreate(&x);
	const int reps = 10;
	float t_f = 0, t_m = 0, t_i = 0;
	for (int r = -2; r < reps; ++r) {
		hipEventRecord(e[0]);
		hipfftExecZ2Z(plan, (hipfftDoubleComplex *) w, (hipfftDoubleComplex *) w, HIPFFT_FORWARD);
		hipEventRecord(e[1]);
		mul_spectrum<<<8192, 256>>>(w, h, n, total);
		hipEventRecord(e[2]);
		hipfftExecZ2Z(plan, (hipfftDoubleComplex *) w, (hipfftDoubleComplex *) w, HIPFFT_BACKWARD);
		hipEventRecord(e[3]);
		hipEventSynchronize(e[3]);
		if (r < 0) continue;
		float a, b, c;
		hipEventElapsedTime(&a, e[0], e[1]);
		hipEventElapsedTime(&b, e[1], e[2]);
		hipEventElapsedTime(&c, e[2], e[3]);
		t_f += a; t_m += b; t_i += c;
	}
	const double gb = (double) total * 16 / 1e9;
	printf("{\"problem\": \"%d x 2^%d complex fp64, in place\", \"work_buffer_bytes\": %zu, \"fwd_ms\": %.3f, \"mul_ms\": %.3f, \"inv_ms\": %.3f, \"total_ms\": %.3f, "
	       "\"buffer_GB\": %.2f, \"min_traffic_GB_if_one_trip_per_call\": %.2f}\n",
	       batch, log2n, ws, t_f / reps, t_m / reps, t_i / reps, (t_f + t_m + t_i) / reps, gb, 6 * gb);
	hipfftDestroy(plan);
	return 0;
}
