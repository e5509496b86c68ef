// kernels_pcm.hip -- the wire formats either side of the chain, on the device.
//
// Replaces, for whole batches of streams resident in HBM,
//   read_buf_<fmt> / write_buf_<fmt>          sampleconv.c:25-149 with the BIT_PERFECT macros of sampleconv.h:35-56
//   clip() + TPDF dither at the sink          dsp.c:673-694, util.h:127-178
// All of it is bit-exact: the conversions are single IEEE operations (scaling by a power of two, nearbyint, a
// saturating compare), and the dither noise is the difference of two Lehmer generators (multipliers 48271 and 16807
// modulo 2^31 - 1, both seeded with 1) advanced once per sample in interleaved order -- sample n of a stream uses
// A^(n+1) mod (2^31 - 1), which every thread reaches by modular exponentiation instead of walking the sequence.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "pcm_params.h"

namespace dspamd {

constexpr uint32_t PM = 0x7fffffffu;      // 2^31 - 1 (util.h:57)

// (a * b) mod (2^31 - 1) the way PM_RAND_R_DEFINE_FUNC folds it (util.h:127-136); values stay in [1, 2^31 - 1]
__device__ __forceinline__ uint32_t pm_mul(uint32_t a, uint32_t b)
{
	const uint64_t p = (uint64_t) a * b;
	uint32_t r = (uint32_t) (p & PM) + (uint32_t) (p >> 31);
	r = (r & PM) + (r >> 31);
	return r;
}

__device__ __forceinline__ uint32_t pm_pow(uint32_t a, uint64_t e)
{
	uint32_t r = 1;
	while (e) {
		if (e & 1) r = pm_mul(r, a);
		a = pm_mul(a, a);
		e >>= 1;
	}
	return r;
}

__device__ __forceinline__ double pcm_load(const void *in, int fmt, long i)
{
	switch (fmt) {
	case PCM_U8: return ((double) static_cast<const uint8_t *>(in)[i] - 128.0) / 128.0;          // U8_TO_SAMPLE
	case PCM_S8: return (double) static_cast<const int8_t *>(in)[i] / 128.0;
	case PCM_S16: return (double) static_cast<const int16_t *>(in)[i] / 32768.0;
	case PCM_S24: {                                                                              // S24_SIGN_EXTEND
		int32_t x = static_cast<const int32_t *>(in)[i];
		x = (x & 0x800000) ? (x | ~0x7fffff) : x;
		return (double) x / 8388608.0;
	}
	case PCM_S32: return (double) static_cast<const int32_t *>(in)[i] / 2147483648.0;
	case PCM_S24_3: {                                                                            // sampleconv.c:108-118
		const uint8_t *b = static_cast<const uint8_t *>(in) + 3 * i;
		int32_t x = (int32_t) b[0] | ((int32_t) b[1] << 8) | ((int32_t) b[2] << 16);
		x = (x & 0x800000) ? (x | ~0x7fffff) : x;
		return (double) x / 8388608.0;
	}
	case PCM_FLOAT: return (double) static_cast<const float *>(in)[i];
	default: return static_cast<const double *>(in)[i];
	}
}

// SAMPLE_TO_<fmt> with BIT_PERFECT = 1 (sampleconv.h:35-41): saturate at the positive end, nearbyint elsewhere
// (round-half-even: the default rounding mode; negative overflow cannot occur after clip())
__device__ __forceinline__ double quant(double x, double scale, double maxv)
{
	const double v = x * scale;
	return (v > maxv) ? maxv : rint(v);
}

__device__ __forceinline__ void pcm_store(void *out, int fmt, long i, double x)
{
	switch (fmt) {
	case PCM_U8: {
		const double v = x * 128.0 + 128.0;
		static_cast<uint8_t *>(out)[i] = (uint8_t) ((v > 255.0) ? 255.0 : rint(v));
		break;
	}
	case PCM_S8: static_cast<int8_t *>(out)[i] = (int8_t) quant(x, 128.0, 127.0); break;
	case PCM_S16: static_cast<int16_t *>(out)[i] = (int16_t) quant(x, 32768.0, 32767.0); break;
	case PCM_S24: static_cast<int32_t *>(out)[i] = (int32_t) quant(x, 8388608.0, 8388607.0); break;
	case PCM_S32: static_cast<int32_t *>(out)[i] = (int32_t) quant(x, 2147483648.0, 2147483647.0); break;
	case PCM_S24_3: {
		const int32_t v = (int32_t) quant(x, 8388608.0, 8388607.0);
		uint8_t *b = static_cast<uint8_t *>(out) + 3 * i;
		b[0] = (uint8_t) (v & 0xff); b[1] = (uint8_t) ((v >> 8) & 0xff); b[2] = (uint8_t) ((v >> 16) & 0xff);
		break;
	}
	case PCM_FLOAT: static_cast<float *>(out)[i] = (float) x; break;
	default: static_cast<double *>(out)[i] = x; break;
	}
}

__global__ __launch_bounds__(256) void pcm_read_kernel(PcmReadParams p)
{
	for (long e = (long) blockIdx.x * blockDim.x + threadIdx.x; e < p.n; e += (long) gridDim.x * blockDim.x)
		p.out[e] = pcm_load(p.in, p.fmt, e);
}

constexpr int PCM_RUN = 16;    // consecutive samples per thread (one modular exponentiation each)

// clip (+ dither) + convert; per stream s: stats[2 s] += clipped samples, stats[2 s + 1] = max(|sample|) (as in dsp.c's
// clip_count / peak), both over the values BEFORE clipping and after dither
__global__ __launch_bounds__(256) void pcm_write_kernel(PcmWriteParams p)
{
	const int s = blockIdx.y;
	const double *in = p.in + (size_t) s * p.in_stride_frames * p.C;
	const long n = p.frames * p.C;
	const size_t out_base = (size_t) s * n;                       // packed [S][frames][C] of the wire format
	unsigned long long clipped = 0;
	double peak = 0.0;
	for (long e0 = ((long) blockIdx.x * blockDim.x + threadIdx.x) * PCM_RUN; e0 < n; e0 += (long) gridDim.x * blockDim.x * PCM_RUN) {
		uint32_t s0 = 0, s1 = 0;
		if (p.dither_mult != 0.0) {
			const uint64_t k = (uint64_t) (p.samples_before + e0);   // samples already drawn for this stream
			s0 = pm_pow(48271u, k);                                  // state after k draws from seed 1 (util.h:151-152)
			s1 = pm_pow(16807u, k);
		}
		const long e1 = (e0 + PCM_RUN < n) ? e0 + PCM_RUN : n;
		for (long e = e0; e < e1; ++e) {
			double x = in[e];
			if (p.dither_mult != 0.0) {
				s0 = pm_mul(s0, 48271u);
				s1 = pm_mul(s1, 16807u);
				x = x + (double) ((int32_t) s0 - (int32_t) s1) * p.dither_mult;     // tpdf_noise, util.h:165-172
			}
			const double a = fabs(x);                                 // clip(), dsp.c:673-682
			peak = fmax(peak, a);
			if (a > 1.0) { ++clipped; x = signbit(x) ? -1.0 : 1.0; }
			pcm_store(p.out, p.fmt, (long) out_base + e, x);
		}
	}
	if (p.stats) {
		if (clipped) atomicAdd(reinterpret_cast<unsigned long long *>(p.stats) + 2 * s, clipped);
		// peak >= 0: its IEEE bit pattern orders like an unsigned integer
		atomicMax(reinterpret_cast<unsigned long long *>(p.stats) + 2 * s + 1, (unsigned long long) __double_as_longlong(peak));
	}
}

void launch_pcm_read(const PcmReadParams &p, hipStream_t st)
{
	if (p.n <= 0) return;
	long blocks = (p.n + 255) / 256;
	if (blocks > 16384) blocks = 16384;
	hipLaunchKernelGGL(pcm_read_kernel, dim3((unsigned) blocks), dim3(256), 0, st, p);
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
