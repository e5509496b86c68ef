// kernels_pcm.hip -- the wire formats either side of the chain, on the device.
//
// Replaces, for whole batches of streams resident in HBM,
//   read_buf_<fmt> / write_buf_<fmt>          sampleconv.c:25-149 with the BIT_PERFECT macros of sampleconv.h:35-56
//   clip() + TPDF dither at the sink          dsp.c:673-694, util.h:127-178
// The per-sample functions live in pcm_device.h (the first / last kernel of a pipeline uses them in its own loads and stores
// where it can: engine.cpp Pipeline::run_wire); these are the stand-alone passes for everything else.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "pcm_params.h"
#include "pcm_device.h"

namespace dspamd {

__global__ __launch_bounds__(256) void pcm_read_kernel(PcmReadParams p)
{
	const long s = blockIdx.y, n = p.frames * p.C;
	const long in0 = s * p.in_stride_frames * p.C;
	double *out = p.out + (size_t) s * p.out_stride_frames * p.C;
	for (long e = (long) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long) gridDim.x * blockDim.x)
		out[e] = pcm_load(p.in, p.fmt, in0 + e);
}

constexpr int PCM_RUN = 16;    // consecutive samples per thread (one modular exponentiation each)

// clip (+ dither) + convert; per stream s: stats[2 s] += clipped samples, stats[2 s + 1] = max(|sample|) (as in dsp.c's
// clip_count / peak), both over the values BEFORE clipping and after dither
__global__ __launch_bounds__(256) void pcm_write_kernel(PcmWriteParams p)
{
	const int s = blockIdx.y;
	const double *in = p.in + (size_t) s * p.in_stride_frames * p.C;
	const long n = p.frames * p.C;
	const size_t out_base = (size_t) s * p.out_stride_frames * p.C;
	unsigned long long clipped = 0;
	double peak = 0.0;
	for (long e0 = ((long) blockIdx.x * blockDim.x + threadIdx.x) * PCM_RUN; e0 < n; e0 += (long) gridDim.x * blockDim.x * PCM_RUN) {
		const bool dither = p.dither_mult != 0.0;
		uint32_t s0 = 0, s1 = 0;
		if (dither) {
			const uint64_t k = (uint64_t) (p.samples_before + e0);   // samples already drawn for this stream
			s0 = pm_pow<0>(k);                                   // state after k draws from seed 1 (util.h:151-152)
			s1 = pm_pow<1>(k);
		}
		const long e1 = (e0 + PCM_RUN < n) ? e0 + PCM_RUN : n;
		for (long e = e0; e < e1; ++e) {
			if (dither) { s0 = pm_mul(s0, PM_A0); s1 = pm_mul(s1, PM_A1); }
			pcm_store(p.out, p.fmt, (long) out_base + e, sink_sample(in[e], dither, s0, s1, p.dither_mult, peak, clipped));
		}
	}
	if (p.stats) {
		if (clipped) atomicAdd(reinterpret_cast<unsigned long long *>(p.stats) + 2 * s, clipped);
		// peak >= 0: its IEEE bit pattern orders like an unsigned integer
		atomicMax(reinterpret_cast<unsigned long long *>(p.stats) + 2 * s + 1, (unsigned long long) __double_as_longlong(peak));
	}
}

void launch_pcm_read(const PcmReadParams &p, int n_streams, hipStream_t st)
{
	const long n = p.frames * p.C;
	if (n <= 0) return;
	long blocks = (n + 255) / 256;
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(pcm_read_kernel, dim3((unsigned) blocks, n_streams), dim3(256), 0, st, p);
}

void launch_pcm_write(const PcmWriteParams &p, int n_streams, hipStream_t st)
{
	const long n = p.frames * p.C;
	if (n <= 0) return;
	long blocks = (n + 256L * PCM_RUN - 1) / (256L * PCM_RUN);
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(pcm_write_kernel, dim3((unsigned) blocks, n_streams), dim3(256), 0, st, p);
}

}  // namespace dspamd
