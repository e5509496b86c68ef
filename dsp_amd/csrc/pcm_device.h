// pcm_device.h -- the wire formats either side of the chain as device functions, shared by the stand-alone conversion
// kernels (kernels_pcm.hip) and by the kernels that do the conversion in their own loads / stores (cascade_rows,
// cascade_kernel, conv_col_inv).
//
// Restates, per sample,
//   read_buf_<fmt> / write_buf_<fmt>          sampleconv.c:25-149 with the BIT_PERFECT macros of sampleconv.h:35-56
//   clip() + TPDF dither at the sink          dsp.c:673-694, util.h:127-178
// All of it is bit-exact: the conversions are single IEEE operations (scaling by a power of two, nearbyint, a
// saturating compare), and the dither noise is the difference of two Lehmer generators (multipliers 48271 and 16807
// modulo 2^31 - 1, both seeded with 1) advanced once per sample in interleaved order -- sample n of a stream uses
// A^(n+1) mod (2^31 - 1), which a thread reaches by modular exponentiation (pm_pow) and leaves by multiplying with a
// precomputed power of A (its stride through the stream), never by walking the sequence.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "pcm_params.h"

namespace dspamd {

constexpr uint32_t PM = 0x7fffffffu;      // 2^31 - 1 (util.h:57)
constexpr uint32_t PM_A0 = 48271u, PM_A1 = 16807u;    // util.h:151-152

// (a * b) mod (2^31 - 1) the way PM_RAND_R_DEFINE_FUNC folds it (util.h:127-136); values stay in [1, 2^31 - 1]
__device__ __forceinline__ uint32_t pm_mul(uint32_t a, uint32_t b)
{
	const uint64_t p = (uint64_t) a * b;
	uint32_t r = (uint32_t) (p & PM) + (uint32_t) (p >> 31);
	r = (r & PM) + (r >> 31);
	return r;
}

// A^e for the two multipliers from byte tables: the generators have period 2^31 - 2 (A^(2^31 - 2) = 1 modulo the prime 2^31 - 1),
// so e is first reduced modulo that, and A^e = T[0][e & 255] T[1][(e >> 8) & 255] T[2][...] T[3][...] with T[k][b] = A^(b 256^k):
// four look-ups and three multiplications where square-and-multiply takes about sixty (measured on K3 with dither at the
// headline shape: 13.7 -> see docs/history.md section 4.6).  The tables are computed by the compiler.
struct PmTables { uint32_t t[2][4][256]; };
constexpr uint32_t pm_mul_c(uint32_t a, uint32_t b)
{
	const uint64_t p = (uint64_t) a * b;
	uint32_t r = (uint32_t) (p & PM) + (uint32_t) (p >> 31);
	r = (r & PM) + (r >> 31);
	return r;
}
constexpr PmTables pm_make_tables()
{
	PmTables T{};
	for (int g = 0; g < 2; ++g) {
		uint32_t base = g ? PM_A1 : PM_A0;
		for (int k = 0; k < 4; ++k) {
			uint32_t v = 1;
			for (int b = 0; b < 256; ++b) { T.t[g][k][b] = v; v = pm_mul_c(v, base); }
			base = v;                                  // base^256
		}
	}
	return T;
}
static __device__ const PmTables PM_TAB = pm_make_tables();

// G = 0: A = 48271, G = 1: A = 16807
template <int G> __device__ __forceinline__ uint32_t pm_pow(uint64_t e);
// a signed exponent (A^-k is the inverse in the group): positions computed relative to a frame that lies before the block
template <int G> __device__ __forceinline__ uint32_t pm_pow_signed(long e)
{
	const long m = (long) (PM - 1);
	long r = e % m;
	if (r < 0) r += m;
	return pm_pow<G>((uint64_t) r);
}
template <int G> __device__ __forceinline__ uint32_t pm_pow(uint64_t e)
{
	const uint32_t r = (uint32_t) (e % (uint64_t) (PM - 1));
	uint32_t v = PM_TAB.t[G][0][r & 255];
	v = pm_mul(v, PM_TAB.t[G][1][(r >> 8) & 255]);
	v = pm_mul(v, PM_TAB.t[G][2][(r >> 16) & 255]);
	return pm_mul(v, PM_TAB.t[G][3][r >> 24]);
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

__device__ __forceinline__ double quant_bf(double x, double scale, double maxv)
{
	const double v = x * scale;
	return (v > maxv) ? maxv : rint(v);               // (a select: both arms are plain values)
}

__device__ __forceinline__ double pcm_from_s16(uint32_t h) { return (double) (int16_t) (uint16_t) h / 32768.0; }

// The same conversions of the 4-byte formats WITHOUT a branch per value: inside the unrolled tile loops of the kernels that
// convert in their own loads / stores a switch on the (wave-uniform) format became a scalar compare-and-branch per sample -- 15 %
// of the cascade's time with float input.  The format is folded into constants once per kernel: the integer formats differ in
// a mask (sign extension of s24) and a power of two, float in which of two results is kept (a bit mask, not a select).
struct WordFormat {
	uint32_t ext_mask;         // s24: the bits S24_SIGN_EXTEND sets when bit 23 is (sampleconv.h), else 0
	double inv_scale, scale, maxv;
	uint32_t float_mask;       // all ones for float
};
__device__ __forceinline__ WordFormat word_format(int fmt)
{
	WordFormat f;
	f.ext_mask = (fmt == PCM_S24) ? 0xff800000u : 0u;
	f.scale = (fmt == PCM_S24) ? 8388608.0 : 2147483648.0;
	f.maxv = (fmt == PCM_S24) ? 8388607.0 : 2147483647.0;
	f.inv_scale = 1.0 / f.scale;                     // exact: a power of two
	f.float_mask = (fmt == PCM_FLOAT) ? 0xffffffffu : 0u;
	return f;
}
__device__ __forceinline__ double pcm_from_word(uint32_t w, const WordFormat &f)
{
	const uint32_t x = w | ((0u - ((w >> 23) & 1u)) & f.ext_mask);      // (x & 0x800000) ? x | ~0x7fffff : x  -- the upper byte stays as it is otherwise
	const double vi = (double) (int32_t) x * f.inv_scale;               // == (double) x / scale
	const double vf = (double) __uint_as_float(w);
	const unsigned long long m = ((unsigned long long) f.float_mask << 32) | f.float_mask;
	return __longlong_as_double((long long) ((((unsigned long long) __double_as_longlong(vf)) & m) | (((unsigned long long) __double_as_longlong(vi)) & ~m)));
}
__device__ __forceinline__ uint32_t pcm_to_word(double x, const WordFormat &f)
{
	const uint32_t wi = (uint32_t) (int32_t) quant_bf(x, f.scale, f.maxv);
	const uint32_t wf = __float_as_uint((float) x);
	return (wf & f.float_mask) | (wi & ~f.float_mask);
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

// sample -> the 16 bits of S16 (the same operations as pcm_store)
__device__ __forceinline__ uint32_t pcm_to_s16(double x) { return (uint32_t) (uint16_t) (int16_t) quant(x, 32768.0, 32767.0); }

// the sink of dsp.c:685-699 for one sample: [+ tpdf_noise (util.h:165-172) from the generator values u0, u1 of this sample],
// clip() (dsp.c:673-682) with its peak / clip_count bookkeeping
__device__ __forceinline__ double sink_sample(double x, bool dither, uint32_t u0, uint32_t u1, double dither_mult, double &peak, unsigned long long &clipped)
{
	// two roundings, as the reference's buf[i] + tpdf_noise(mult) has them (util.h:165-172 returns the rounded product): hipcc
	// fuses a * b + c wherever it is allowed to, and whether it does depends on the calling kernel -- the peak it reports would
	// differ in the last bit from one kernel to the next
#pragma clang fp contract(off)
	if (dither) {
		const double noise = (double) ((int32_t) u0 - (int32_t) u1) * dither_mult;
		x = x + noise;
	}
	const double a = fabs(x);
	peak = fmax(peak, a);
	if (a > 1.0) { ++clipped; x = signbit(x) ? -1.0 : 1.0; }
	return x;
}

// A thread that walks a stream's samples at a constant stride (grid-stride loops): the generator values of its first sample by
// modular exponentiation, every further one by one multiplication with A^stride
struct SinkWalk {
	uint32_t u0 = 0, u1 = 0, j0 = 1, j1 = 1;
	double peak = 0.0;
	unsigned long long clipped = 0;
	bool started = false;
	// sample n (0-based in this call's destination, interleaved order) -> what goes on the wire; consecutive calls of one thread
	// must be `stride` samples apart
	__device__ __forceinline__ double next(const WireSink &k, long n, long stride, double x)
	{
		const bool dither = k.dither_mult != 0.0;
		if (dither) {
			if (!started) {
				const uint64_t e = (uint64_t) (k.samples_before + n) + 1;
				u0 = pm_pow<0>(e); u1 = pm_pow<1>(e);
				j0 = pm_pow<0>((uint64_t) stride); j1 = pm_pow<1>((uint64_t) stride);
				started = true;
			}
			else { u0 = pm_mul(u0, j0); u1 = pm_mul(u1, j1); }
		}
		return sink_sample(x, dither, u0, u1, k.dither_mult, peak, clipped);
	}
};

// per-stream statistics of the sink: stats[2 s] += clipped samples (64-bit count), stats[2 s + 1] = max(|sample|).
// Every thread of the WORKGROUP works on the same stream and all of them get here: waves by shuffles, the workgroup through
// LDS, then at most one pair of global atomics -- and the maximum only when it would change something (a million waves
// hitting 256 addresses cost K3 a millisecond at the headline shape).
__device__ __forceinline__ void sink_stats_block(double *stats, long s, double peak, unsigned long long clipped)
{
	__shared__ unsigned long long red[2];
	if (threadIdx.x == 0) { red[0] = 0; red[1] = 0; }
	__syncthreads();
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		peak = fmax(peak, __shfl_xor(peak, d, 64));
		clipped += (unsigned long long) __shfl_xor((long long) clipped, d, 64);
	}
	// peak >= 0: its IEEE bit pattern orders like an unsigned integer
	const unsigned long long pk = (unsigned long long) __double_as_longlong(peak);
	if ((threadIdx.x & 63) == 0) {
		if (clipped) atomicAdd(&red[0], clipped);
		if (pk) atomicMax(&red[1], pk);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long *g = reinterpret_cast<unsigned long long *>(stats) + 2 * s;
		if (red[0]) atomicAdd(g, red[0]);
		if (red[1] > __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(g + 1, red[1]);
	}
}

// the same for a kernel whose waves may work on different streams (rows shorter than a workgroup): one reduction per wave when all its
// lanes share the stream, else each lane for itself; every lane of the wave gets here (`mine` = this lane has a stream at all)
__device__ __forceinline__ void sink_stats_wave(double *stats, long s, bool mine, double peak, unsigned long long clipped)
{
	const int s_first = __builtin_amdgcn_readfirstlane((int) s);
	const bool uniform = __builtin_amdgcn_ballot_w64(mine && (int) s != s_first) == 0;
	unsigned long long pk = (unsigned long long) __double_as_longlong(mine ? peak : 0.0);
	if (!mine) clipped = 0;
	bool writer = mine;
	if (uniform) {
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1) {
			const unsigned long long o = (unsigned long long) __shfl_xor((long long) pk, d, 64);
			pk = (o > pk) ? o : pk;
			clipped += (unsigned long long) __shfl_xor((long long) clipped, d, 64);
		}
		writer = (threadIdx.x & 63) == 0;
		s = s_first;
	}
	if (writer && (clipped || pk)) {
		unsigned long long *g = reinterpret_cast<unsigned long long *>(stats) + 2 * s;
		if (clipped) atomicAdd(g, clipped);
		if (pk > __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(g + 1, pk);
	}
}

}  // namespace dspamd
