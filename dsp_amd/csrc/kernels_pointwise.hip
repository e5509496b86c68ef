// kernels_pointwise.hip -- the bit-exact class (remix, integer delay / align, slab copy) and the device-side
// bench endpoints (sgen sine source, digest sink, copy probe).
//
//   remix_kernel  : remix_effect_run_{1a,4,generic}   remix.c:39-101  (ascending-input-channel sums from 0.0)
//                   + weighted rows: st2ms / ms2st (st2ms.c:28-54), the two ends of a crossfeed (crossfeed.c:41-46)
//   delay_kernel  : align_channel_run                 align.c:35-44   (per-channel delay of len frames, state carried)
//   sgen_kernel   : sgen_run_generator (sine)         sgen.c:55-67
//   digest_kernel : the `stats` quantities            stats.c:47-76
// All arithmetic that must match the CPU bit for bit is written with plain operators under `#pragma clang fp contract(off)`:
// hipcc fuses a * b + c by default, and it does so THROUGH __dmul_rn / __dadd_rn (inlined helpers keep the translation
// unit's contraction flag) -- only the pragma, applied to operators in its own scope, guarantees one rounding per operation.
// remix_kernel and delay_kernel also speak the wire formats (pcm_device.h) when they are the first / last kernel of a pipeline
// run from wire format to wire format (engine.cpp Pipeline::run_wire): every format, element by element.
#include <hip/hip_runtime.h>
#include "kparams.h"
#include "pcm_device.h"

namespace dspamd {

__global__ __launch_bounds__(256) void remix_kernel(RemixParams p)
{
#pragma clang fp contract(off)
	const int s = blockIdx.y;
	const long n = p.frames * p.Cout;
	const long in0 = (long) s * p.in_stride_frames * p.Cin, out0 = (long) s * p.out_stride_frames * p.Cout;
	const double *in = p.in + in0;
	double *out = p.out + out0;
	const bool wire_in = p.in_fmt != PCM_DOUBLE;
	const long stride = (long) gridDim.x * blockDim.x;
	SinkWalk walk;
	auto x = [&](long t, int c) { return wire_in ? pcm_load(p.in, p.in_fmt, in0 + t * p.Cin + c) : in[t * p.Cin + c]; };
	for (long e = (long) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
		const long t = e / p.Cout;
		const int k = (int) (e - t * p.Cout);
		const int *idx = p.idx + (size_t) k * p.max_n;
		double acc = 0.0;
		if (p.w) {
			// weighted form (st2ms.c:34-38, crossfeed.c:41-46): the first product starts the sum, every operation rounds once
			const double *w = p.w + (size_t) k * p.max_n;
			if (idx[0] >= 0) acc = x(t, idx[0]) * w[0];
			for (int j = 1; j < p.max_n; ++j) {
				const int c = idx[j];
				if (c < 0) break;
				const double prod = x(t, c) * w[j];
				acc = acc + prod;
			}
			if (p.post) acc = acc * p.post[k];
		}
		else {
			for (int j = 0; j < p.max_n; ++j) {
				const int c = idx[j];
				if (c < 0) break;
				acc = acc + x(t, c);
			}
		}
		if (p.sink.on) pcm_store(p.out, p.sink.fmt, out0 + e, walk.next(p.sink, e, stride, acc));
		else out[e] = acc;
	}
	if (p.sink.on && p.sink.stats) sink_stats_block(p.sink.stats, s, walk.peak, walk.clipped);
}

void launch_remix(const RemixParams &p, int n_streams, hipStream_t stream)
{
	const long n = p.frames * p.Cout;
	if (n <= 0) return;
	long blocks = (n + 255) / 256;
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(remix_kernel, dim3((unsigned) blocks, n_streams), dim3(256), 0, stream, p);
}

// out[t] = x[t - len]; the ring holds the last len input frames.  ring is double-buffered by the host
// (read `ring`, write `ring + ring_alt_off`) so that no slot is read and written by different threads.
struct DelayKArgs {
	DelayParams p;
	long ring_alt_off;   // offset (doubles) from the read ring to the write ring
	long skip;           // leading output frames dropped (end-of-chain discard, align.c:53-62)
	long max_len;
};

__global__ __launch_bounds__(256) void delay_kernel(DelayKArgs a)
{
	const DelayParams &p = a.p;
	const int s = blockIdx.y;
	const long span = (p.frames > a.max_len) ? p.frames : a.max_len;
	const long n = span * p.C;
	const long in0 = (long) s * p.in_stride_frames * p.C, out0 = (long) s * p.out_stride_frames * p.C;
	const double *in = p.in + in0;
	double *out = p.out + out0;
	const double *rd = p.ring + (size_t) s * p.ring_per_stream;
	double *wr = p.ring + (size_t) s * p.ring_per_stream + a.ring_alt_off;
	const bool wire_in = p.in_fmt != PCM_DOUBLE;
	const long stride = (long) gridDim.x * blockDim.x;
	SinkWalk walk;
	auto x = [&](long i) { return wire_in ? pcm_load(p.in, p.in_fmt, in0 + i) : in[i]; };
	for (long e = (long) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
		const long t = e / p.C;
		const int c = (int) (e - t * p.C);
		const long len = p.len[c];
		if (t < p.frames) {
			double v;
			if (len == 0) v = x(t * p.C + c);
			else if (t >= len) v = x((t - len) * p.C + c);
			else v = rd[p.ring_off[c] + (p.pos + t) % len];
			if (t >= a.skip) {
				// (the outputs of a thread are its iterations from the first with t >= skip on: `stride` samples apart)
				const long o = (t - a.skip) * p.C + c;
				if (p.sink.on) pcm_store(p.out, p.sink.fmt, out0 + o, walk.next(p.sink, o, stride, v));
				else out[o] = v;
			}
		}
		if (len > 0 && t < span) {
			if (t < p.frames) {
				if (t >= p.frames - len) wr[p.ring_off[c] + (p.pos + t) % len] = x(t * p.C + c);
			}
			else if (t < len) {   // slot not overwritten this block: carry it over
				const long slot = (p.pos + t) % len;
				wr[p.ring_off[c] + slot] = rd[p.ring_off[c] + slot];
			}
		}
	}
	if (p.sink.on && p.sink.stats) sink_stats_block(p.sink.stats, s, walk.peak, walk.clipped);
}

void launch_delay_ex(const DelayParams &p, long ring_alt_off, long skip, long max_len, int n_streams, hipStream_t stream)
{
	DelayKArgs a{ p, ring_alt_off, skip, max_len };
	const long span = (p.frames > max_len) ? p.frames : max_len;
	const long n = span * p.C;
	if (n <= 0) return;
	long blocks = (n + 255) / 256;
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(delay_kernel, dim3((unsigned) blocks, n_streams), dim3(256), 0, stream, a);
}

__global__ __launch_bounds__(256) void copy_slab_kernel(const double *in, long in_stride, double *out, long out_stride, long frames, long skip, int C)
{
	const int s = blockIdx.y;
	const double *src = in + ((size_t) s * in_stride + skip) * C;
	double *dst = out + (size_t) s * out_stride * C;
	const long n = frames * C;
	for (long e = (long) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long) gridDim.x * blockDim.x)
		dst[e] = src[e];
}

void launch_copy_slab(const double *in, long in_stride, double *out, long out_stride, long frames, long skip, int C, int n_streams, hipStream_t stream)
{
	const long n = frames * C;
	if (n <= 0) return;
	long blocks = (n + 255) / 256;
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(copy_slab_kernel, dim3((unsigned) blocks, n_streams), dim3(256), 0, stream, in, in_stride, out, out_stride, frames, skip, C);
}

// stream s: sin(2 pi f_s t), t = (pos0 + frame) / fs, the same value on every channel (sgen.c:55-67).
// Device libm differs from the host's by ulps: this source is for throughput runs; parity runs feed
// byte-identical host-generated input to both sides.
__global__ __launch_bounds__(256) void sgen_kernel(double *buf, long frames, int C, int fs, double freq0, double dfreq, long pos0)
{
	const int s = blockIdx.y;
	const double w = (freq0 + s * dfreq) * (2.0 * M_PI);
	double *dst = buf + (size_t) s * frames * C;
	const long n = frames * C;
	for (long e = (long) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long) gridDim.x * blockDim.x) {
		const long t = e / C;
		dst[e] = sin(w * ((double) (pos0 + t) / fs));
	}
}

// the other two forms of sgen_run_generator: a sine sweep freq0 -> freq1 over total_frames (sgen.c:60-62 with v of :163:
// sin(w0 / v (exp(t v) - 1)), v = log(w1 / w0) / (total_frames / fs)) and a unit impulse at frame `offset` (sgen.c:46-52).
// kind 1 = sweep, 2 = delta; stream s uses freq0 + s dfreq (and the same ratio freq1 / freq0), offset + s doffset
__global__ __launch_bounds__(256) void sgen_kernel2(double *buf, long frames, int C, int fs, int kind, double freq0, double freq1, double dfreq, long total_frames, long offset, long doffset, long pos0)
{
	const int s = blockIdx.y;
	const double w0 = (freq0 + s * dfreq) * (2.0 * M_PI), w1 = (freq1 + s * dfreq * (freq1 / freq0)) * (2.0 * M_PI);
	const double v = (kind == 1 && total_frames > 0 && w0 != w1) ? log(w1 / w0) / ((double) total_frames / fs) : 0.0;
	const long off = offset + s * doffset;
	double *dst = buf + (size_t) s * frames * C;
	const long n = frames * C;
	for (long e = (long) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long) gridDim.x * blockDim.x) {
		const long fr = pos0 + e / C;
		if (kind == 2) { dst[e] = (fr == off) ? 1.0 : 0.0; continue; }
		const double t = (double) fr / fs;
		dst[e] = (v != 0.0) ? sin(w0 / v * (exp(t * v) - 1.0)) : sin(w0 * t);
	}
}

void launch_sgen(double *buf, int n_streams, long frames, int channels, int fs, int kind, double freq0, double freq1, double dfreq, long total_frames, long offset, long doffset, long pos0, hipStream_t stream)
{
	const long n = frames * channels;
	if (n <= 0) return;
	long blocks = (n + 255) / 256;
	if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(sgen_kernel2, dim3((unsigned) blocks, n_streams), dim3(256), 0, stream, buf, frames, channels, fs, kind, freq0, freq1, dfreq, total_frames, offset, doffset, pos0);
}

void launch_sgen_sine(double *buf, int n_streams, long frames, int channels, int fs, double freq0, double dfreq, long pos0, hipStream_t stream)
{
	const long n = frames * channels;
	if (n <= 0) return;
	long blocks = (n + 255) / 256;
	if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(sgen_kernel, dim3((unsigned) blocks, n_streams), dim3(256), 0, stream, buf, frames, channels, fs, freq0, dfreq, pos0);
}

// per stream: sum, sum of squares, peak |x|  ->  out[s][3]   (one workgroup per stream; deterministic order)
__global__ __launch_bounds__(256) void digest_kernel(const double *buf, long frames, long stride, int C, double *out)
{
	__shared__ double sh[3][256];
	const int s = blockIdx.x;
	const double *src = buf + (size_t) s * stride * C;
	const long n = frames * C;
	double a = 0.0, b = 0.0, m = 0.0;
	for (long e = threadIdx.x; e < n; e += blockDim.x) {
		const double v = src[e];
		a += v;
		b += v * v;
		m = fmax(m, fabs(v));
	}
	sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b; sh[2][threadIdx.x] = m;
	__syncthreads();
	for (int k = 128; k > 0; k >>= 1) {
		if ((int) threadIdx.x < k) {
			sh[0][threadIdx.x] += sh[0][threadIdx.x + k];
			sh[1][threadIdx.x] += sh[1][threadIdx.x + k];
			sh[2][threadIdx.x] = fmax(sh[2][threadIdx.x], sh[2][threadIdx.x + k]);
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) { out[3*s] = sh[0][0]; out[3*s + 1] = sh[1][0]; out[3*s + 2] = sh[2][0]; }
}

void launch_digest(const double *buf, int n_streams, long frames, long stride, int channels, double *out, hipStream_t stream)
{
	hipLaunchKernelGGL(digest_kernel, dim3(n_streams), dim3(256), 0, stream, buf, frames, stride, channels, out);
}

// 16 B per lane streaming copy: the measured HBM ceiling quoted next to the 8 TB/s spec.  One 4-KiB tile per workgroup,
// one access in flight per lane: the fastest of the variants in scripts/ubench/hbmprobe.hip on this chip (6.2-6.3 TB/s;
// round 1's 2048-workgroup grid-stride loop read 4.7-4.9, and more accesses in flight per lane read LESS: 16 per lane 5.3-5.6)
__global__ __launch_bounds__(256) void copy_probe_kernel(const double2 *__restrict__ src, double2 *__restrict__ dst, size_t n)
{
	const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
	if (i < n) dst[i] = src[i];
}

void launch_copy_probe(const void *src, void *dst, size_t bytes, hipStream_t stream)
{
	const size_t n = bytes / 16;
	if (!n) return;
	hipLaunchKernelGGL(copy_probe_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, (const double2 *) src, (double2 *) dst, n);
}

}  // namespace dspamd
