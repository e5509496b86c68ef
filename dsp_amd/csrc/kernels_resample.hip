// kernels_resample.hip -- rational resampler as a polyphase dot-product kernel.
//
// Replaces resample_effect_run / resample_effect_drain2 (resample.c:89-152, 163-188).  The reference works in
// the FFT domain (r2c of 2*in_len samples, image/fold multiply by the spectrum of a windowed-sinc prototype,
// c2r of 2*out_len samples, overlap-add).  In the time domain that is exactly (checked to
// 1e-16 / 6e-15 RMS against the reference for integer and general n/d ratios):
//     full[k] = A * sum_q x[q] * s_a( (k d - q n) * os / min(n, d) ),       y[m] = full[m + out_delay]
// with s_a the analytic prototype (norm_sinc * Albrecht window, resample.c:52-87) evaluated at fractional
// positions.  For output k only the phase p = (k d) mod n and q0 = floor(k d / n) matter, so the host tabulates
// tab[j][p] = A s_a((p + j n) os / min(n,d)) once (n phases x J taps) and the kernel is a batched dot product
// over the J most recent input frames.  Not a dense GEMM (each output row uses a different phase column and a
// sliding input window), so no MFMA: fp64 FMAs fed from LDS.
//
// Work split: workgroup = (stream, tile of KT outputs); the input span of the tile is staged in LDS once
// (all channels), every thread owns one output frame and CPT channels of it.
#include <hip/hip_runtime.h>
#include "resample_params.h"
#include "kparams.h"
#include "pcm_device.h"

namespace dspamd {

template <int CPT>
__global__ __launch_bounds__(256) void resample_kernel(ResampleParams p)
{
	extern __shared__ __attribute__((aligned(16))) double xs[];   // [span][C]
	const int s = blockIdx.y;
	const long m0 = p.m_first + (long) blockIdx.x * p.KT;
	const int nk = (int) min((long) p.KT, p.m_first + p.m_count - m0);
	if (nk <= 0) return;
	const int C = p.C;
	// input span needed by outputs [m0, m0 + nk)
	const long k_lo = m0 + p.out_delay, k_hi = m0 + nk - 1 + p.out_delay;
	const long q_hi = (k_hi * p.d) / p.n;
	const long q_lo = (k_lo * p.d) / p.n - (p.J - 1);
	const int span = (int) (q_hi - q_lo + 1);
	const double *ring = p.ring + (size_t) s * p.ring_len * C;
	for (int e = threadIdx.x; e < span * C; e += blockDim.x) {
		const long q = q_lo + e / C;
		const int c = e % C;
		xs[e] = (q >= 0 && q < p.q_total) ? ring[((q & p.ring_mask)) * C + c] : 0.0;
	}
	__syncthreads();
	const int groups = (C + CPT - 1) / CPT;
	double *out = p.out + (size_t) s * p.out_stride_frames * C;
	double sink_peak = 0.0;
	unsigned long long sink_clipped = 0;
	for (int w = threadIdx.x; w < nk * groups; w += blockDim.x) {
		const int kl = w / groups, cg = (w % groups) * CPT;
		const long k = m0 + kl + p.out_delay;
		const long q0 = (k * p.d) / p.n;
		const int ph = (int) (k * p.d - q0 * p.n);
		const double *x = xs + (size_t) (q0 - q_lo) * C + cg;
		const double *tab = p.tab + ph;
		double acc[CPT];
#pragma unroll
		for (int i = 0; i < CPT; ++i) acc[i] = 0.0;
		for (int j = 0; j < p.J; ++j) {
			const double t = tab[(size_t) j * p.n];
#pragma unroll
			for (int i = 0; i < CPT; ++i)
				if (cg + i < C) acc[i] = fma(t, x[i], acc[i]);
			x -= C;
		}
		if (p.sink.on) {
			// the last kernel of a pipeline run in wire formats (dsp.c:685-699): every sample finds its place in the dither sequences by
			// itself (4 table look-ups and 3 modular multiplications per generator: nothing beside the J taps it took to compute)
#pragma unroll
			for (int i = 0; i < CPT; ++i) {
				if (cg + i >= C) continue;
				const long idx = (m0 - p.m_first + kl + p.out_frame0) * C + cg + i;
				const bool dither = p.sink.dither_mult != 0.0;
				const uint64_t nn = (uint64_t) (p.sink.samples_before + idx) + 1;
				const double y = sink_sample(acc[i], dither, dither ? pm_pow<0>(nn) : 0u, dither ? pm_pow<1>(nn) : 0u, p.sink.dither_mult, sink_peak, sink_clipped);
				pcm_store(reinterpret_cast<char *>(p.out) + (size_t) s * p.out_stride_frames * C * p.sink_bs, p.sink.fmt, idx, y);
			}
			continue;
		}
#pragma unroll
		for (int i = 0; i < CPT; ++i)
			if (cg + i < C) out[(m0 - p.m_first + kl + p.out_frame0) * C + cg + i] = acc[i];
	}
	if (p.sink.on && p.sink.stats) sink_stats_block(p.sink.stats, s, sink_peak, sink_clipped);
}

// append `frames` interleaved frames to the per-stream history ring
__global__ __launch_bounds__(256) void resample_push_kernel(const double *in, long in_stride, double *ring, long ring_len, long ring_mask, long pos, long frames, int C)
{
	const int s = blockIdx.y;
	const double *src = in + (size_t) s * in_stride * C;
	double *dst = ring + (size_t) s * ring_len * C;
	const long n = frames * C;
	for (long e = (long) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long) gridDim.x * blockDim.x) {
		const long t = e / C;
		dst[((pos + t) & ring_mask) * C + (e - t * C)] = src[e];
	}
}

// ---------------------------------------------------------------------------------------------------------
// Rational n/d resampling as a dense GEMM on the fp64 matrix cores.
//
// Group the full-rate outputs in blocks of NB = n g consecutive outputs (g = 1 unless n is small); block i uses
// the inputs x[DB i + u], u in [-(J-1), DB-1], DB = d g.  With r the position inside the block
//     full[NB i + r] = sum_j tab[j][(r d) mod n] x[DB i + floor(r d / n) - j]  =  sum_u X[i][u] G[u][r],
//     X[i][u] = x[DB i + u],      G[u][r] = tab[floor(r d / n) - u][(r d) mod n]   (0 outside the J taps)
// i.e. Y = X G with a DENSE K x NB matrix G (K = J + DB - 1; 80 % non-zero for 147/160) that is the same for every
// block, channel and stream: a genuine GEMM (M = blocks x channels, N = NB, K = J + DB - 1), unlike the per-output
// dot products of resample_kernel (a gathered table read per tap).  v_mfma_f64_16x16x4_f64: A = X rows read from
// the LDS copy of the input span (Hankel structure: row i+1 is row i shifted by DB frames, so the span is staged
// once), B = G from L2 (<= 1 MB, shared by all workgroups), C in registers.
//
// Workgroup = 4 waves = 64 rows (IB blocks x CP channels, CP = channels padded to a power of two) x all NB columns;
// wave (wr, wp) owns 2 row tiles x half of the column tiles.  LDS frame f of channel c sits at
// f CP + c + 16 (f >> 5) doubles (padding keeps the two blocks of a row tile on different banks).
typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int RSG_ROWS = 64;

__device__ __forceinline__ int rsg_lds_index(int f, int c, int log2cp) { return (f << log2cp) + c + ((f >> 5) << 4); }

template <int PT>
__global__ __launch_bounds__(256) void resample_gemm_kernel(ResampleGemmParams p)
{
	extern __shared__ __attribute__((aligned(16))) double xs[];
	const int s = blockIdx.y;
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int wr = w >> 1, wp = w & 1;
	const int CP = 1 << p.log2cp, IB = RSG_ROWS >> p.log2cp;
	const long i0 = p.i_first + (long) blockIdx.x * IB;            // first block of this workgroup
	// ---- stage the input span: frames q in [DB i0 - (J-1), DB (i0 + IB - 1) + DB - 1] ----
	const long q_lo = (long) p.DB * i0 - (p.J - 1);
	const int span = p.DB * IB + p.J - 1;
	const double *ring = p.ring + (size_t) s * p.ring_len * p.C;
	for (int e = tid; e < (span + 16) * CP; e += 256) {      // + 16: the K padding / prefetch read a few frames past the span (x 0 in G)
		const int f = e >> p.log2cp, c = e & (CP - 1);
		const long q = q_lo + f;
		double v = 0.0;
		if (c < p.C && q >= 0 && q < p.q_total) v = ring[(q & p.ring_mask) * p.C + c];
		xs[rsg_lds_index(f, c, p.log2cp)] = v;
	}
	__syncthreads();
	// ---- GEMM: this wave's 2 row tiles x PT column tiles ----
	v4d acc[2][PT];
#pragma unroll
	for (int a = 0; a < 2; ++a)
#pragma unroll
		for (int b = 0; b < PT; ++b) acc[a][b] = (v4d) { 0.0, 0.0, 0.0, 0.0 };
	// A fragment: lane (row r = lane & 15, k = lane >> 4) of row tile t: row R = 16 t + r -> block il = R / CP, channel c = R % CP;
	// element X[il][uu + k] = frame DB il + uu + k of the span
	int abase[2];
#pragma unroll
	for (int a = 0; a < 2; ++a) {
		const int R = 16 * (2 * wr + a) + (lane & 15);
		abase[a] = p.DB * (R >> p.log2cp) + (lane >> 4);            // frame index at uu = 0
	}
	const int ac = (16 * (2 * wr) + (lane & 15)) & (CP - 1);        // channel of this lane's rows (same for both tiles: 16 % CP == 0)
	const int ptiles = (p.NB + 15) >> 4;
	const int pt0 = wp * PT;                                        // first column tile of this wave
	const double *G = p.G + (size_t) (lane >> 4) * p.Npad + (lane & 15);
	// software pipeline with two register sets (Kpad is a multiple of 8, G is padded to 2 PT column tiles and has a
	// zero row block after the last step, so no load needs a guard): the fragments of the next step are requested
	// before the MFMAs of the current one -- with one wave per SIMD nothing else hides the L2 latency of the B loads
	const double *Gw = G + 16 * pt0;
	double b0[PT], b1[PT], a0[2], a1[2];
#pragma unroll
	for (int b = 0; b < PT; ++b) b0[b] = Gw[16 * b];
#pragma unroll
	for (int a = 0; a < 2; ++a) a0[a] = xs[rsg_lds_index(abase[a], ac, p.log2cp)];
	for (int uu = 0; uu < p.Kpad; uu += 8) {
#pragma unroll
		for (int b = 0; b < PT; ++b) b1[b] = Gw[(size_t) (uu + 4) * p.Npad + 16 * b];
#pragma unroll
		for (int a = 0; a < 2; ++a) a1[a] = xs[rsg_lds_index(abase[a] + uu + 4, ac, p.log2cp)];
#pragma unroll
		for (int a = 0; a < 2; ++a)
#pragma unroll
			for (int b = 0; b < PT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[a], b0[b], acc[a][b], 0, 0, 0);
#pragma unroll
		for (int b = 0; b < PT; ++b) b0[b] = Gw[(size_t) (uu + 8) * p.Npad + 16 * b];
#pragma unroll
		for (int a = 0; a < 2; ++a) a0[a] = xs[rsg_lds_index(abase[a] + uu + 8, ac, p.log2cp)];
#pragma unroll
		for (int a = 0; a < 2; ++a)
#pragma unroll
			for (int b = 0; b < PT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[a], b1[b], acc[a][b], 0, 0, 0);
	}
	// ---- store: C[row = (lane >> 4) + 4 reg][col = lane & 15] ----
	double *out = p.out + (size_t) s * p.out_stride_frames * p.C;
	char *wout = reinterpret_cast<char *>(p.out) + (size_t) s * p.out_stride_frames * p.C * p.sink_bs;     // (the sink's samples)
	const bool dither = p.sink.on && p.sink.dither_mult != 0.0;
	double sink_peak = 0.0;
	unsigned long long sink_clipped = 0;
	// one 16 x 16 tile: C[row = (lane >> 4) + 4 reg][col = lane & 15]
	auto store_tile = [&](int a, int b, const v4d tile, bool to_sink) {
		const int r = 16 * (pt0 + b) + (lane & 15);             // position inside the block
		if (pt0 + b >= ptiles || r >= p.NB) return;
#pragma unroll
		for (int reg = 0; reg < 4; ++reg) {
			const int R = 16 * (2 * wr + a) + (lane >> 4) + 4 * reg;
			const int il = R >> p.log2cp, c = R & (CP - 1);
			const long m = (long) p.NB * (i0 + il) + r - p.out_delay - p.m_first;   // visible output frame of this call
			if (c < p.C && m >= 0 && m < p.m_count) {
				const long idx = (p.out_frame0 + m) * p.C + c;
				if (to_sink) {
					// (every sample finds its place in the dither sequences by itself: see resample_kernel)
					const uint64_t nn = (uint64_t) (p.sink.samples_before + idx) + 1;
					const double y = sink_sample(tile[reg], dither, dither ? pm_pow<0>(nn) : 0u, dither ? pm_pow<1>(nn) : 0u, p.sink.dither_mult, sink_peak, sink_clipped);
					pcm_store(wout, p.sink.fmt, idx, y);
				}
				else out[idx] = tile[reg];
			}
		}
	};
	if (!p.sink.on) {
#pragma unroll
		for (int a = 0; a < 2; ++a)
#pragma unroll
			for (int b = 0; b < PT; ++b) store_tile(a, b, acc[a][b], false);
	}
	else {
		// the sink's code (dither sequences, clipping, five wire formats) ONCE, in a loop over the tiles that picks its accumulator by a wave-uniform
		// select: unrolled 2 x PT x 4 times it is too large for the unroller at PT = 6, which then left the loop over a / b rolled and the accumulators
		// in scratch -- 416 bytes per lane until round 6
#pragma unroll 1
		for (int ab = 0; ab < 2 * PT; ++ab) {
			v4d tile = acc[0][0];
#pragma unroll
			for (int k = 1; k < 2 * PT; ++k) if (ab == k) tile = acc[k / PT][k % PT];
			store_tile(ab / PT, ab % PT, tile, true);
		}
	}
	if (p.sink.on && p.sink.stats) sink_stats_block(p.sink.stats, s, sink_peak, sink_clipped);
}

void launch_resample_gemm(const ResampleGemmParams &p, int n_streams, hipStream_t st)
{
	if (p.m_count <= 0) return;
	const int IB = RSG_ROWS >> p.log2cp;
	// blocks covering full-rate indices [m_first + out_delay, m_first + m_count + out_delay)
	const long i_last = (p.m_first + p.m_count - 1 + p.out_delay) / p.NB;
	const long nblk = i_last - p.i_first + 1;
	const long tiles = (nblk + IB - 1) / IB;
	const size_t lds = resample_gemm_lds_bytes(p.DB, p.J, p.log2cp);
	const int PT = p.Npad >> 5;
	dim3 grid((unsigned) tiles, n_streams), block(256);
#define RSG_LAUNCH(N_)                                                                                                       \
	{                                                                                                                        \
		grant_dynamic_lds(reinterpret_cast<const void *>(resample_gemm_kernel<N_>), lds);                                  \
		hipLaunchKernelGGL(resample_gemm_kernel<N_>, grid, block, lds, st, p);                                                \
	}
	if (PT <= 1) RSG_LAUNCH(1) else if (PT == 2) RSG_LAUNCH(2) else if (PT == 3) RSG_LAUNCH(3) else if (PT == 4) RSG_LAUNCH(4) else if (PT == 5) RSG_LAUNCH(5) else RSG_LAUNCH(6)
#undef RSG_LAUNCH
}

size_t resample_gemm_lds_bytes(int DB, int J, int log2cp)
{
	const int IB = RSG_ROWS >> log2cp;
	const long span = (long) DB * IB + J - 1 + 16;
	return (size_t) ((span << log2cp) + ((span >> 5) + 1) * 16) * sizeof(double);
}

void launch_resample(const ResampleParams &p, int n_streams, hipStream_t st)
{
	if (p.m_count <= 0) return;
	const long tiles = (p.m_count + p.KT - 1) / p.KT;
	const long span_max = ((long) p.KT * p.d) / p.n + p.J + 2;
	const size_t lds = (size_t) span_max * p.C * sizeof(double);
	grant_dynamic_lds(reinterpret_cast<const void *>(resample_kernel<4>), lds);
	hipLaunchKernelGGL(resample_kernel<4>, dim3((unsigned) tiles, n_streams), dim3(256), lds, st, p);
}

void launch_resample_push(const double *in, long in_stride, double *ring, long ring_len, long pos, long frames, int C, int n_streams, hipStream_t st)
{
	const long n = frames * C;
	if (n <= 0) return;
	long blocks = (n + 255) / 256;
	if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(resample_push_kernel, dim3((unsigned) blocks, n_streams), dim3(256), 0, st, in, in_stride, ring, ring_len, ring_len - 1, pos, frames, C);
}

}  // namespace dspamd
