// kernels_fft.hip -- overlap-save FFT convolution for fir / fir_p / hilbert / zita-equivalent, hand-written for gfx950.
//
// Replaces the reference's FFTW-based paths
//   fir_effect_run                      fir.c:109-149      (one 2*len FFT per len samples)
//   fir_p_effect_run + fft_part_group_compute   fir_p.c:64-181   (non-uniform partitions + frequency-domain delay line)
// by their common mathematical content: y[n] = sum_k h[k] x[n-k] per channel, streamed.
//
// Design (DESIGN.md "FFT convolver"):
//  * Two real channels that share one real filter ride one COMPLEX transform: z = x_a + i x_b,
//    IFFT(FFT(z) H) = (x_a * h) + i (x_b * h) because h is real -- no real-FFT split step, no wasted half spectrum.
//  * One big transform per block instead of the reference's many small partitions: the frequency-domain
//    delay line of fir_p re-reads ~16 B x (taps / partition) per sample; a single N-point overlap-save
//    block touches each sample O(1) times per pass.
//  * N = N1 x N2 four-step decomposition with the two middle passes fused, so a block makes three trips
//    through HBM instead of the five a library FFT -> multiply -> library IFFT sequence needs:
//      K1 conv_col_fwd : gather z from the planar rings, FFT over n1 (stride N2), twiddle  -> W
//      K2 conv_row     : FFT over n2 (contiguous), x H, IFFT over k2, conj twiddle         -> W (in place)
//      K3 conv_col_inv : IFFT over k1, scatter the valid outputs to the interleaved slab
//  * Everything fp64 (the reference is fp64 end to end); twiddles come from tables built in extended
//    precision on the host, the big inter-pass twiddle w_N^(n2 k1) from a two-level table (one complex
//    multiply) so that no 2 MB table is streamed.
//  * LDS-resident Stockham radix-4 passes (radix-2 tail), 64-wide waves, 16-byte ds accesses; column
//    tiles are 16 points wide so every global access is a 256 B run.
#include <hip/hip_runtime.h>
#include "kparams.h"
#include "fft_params.h"

namespace dspamd {

typedef double2 cplx;

__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cmulc(cplx a, cplx b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }

constexpr int NT = 256;   // threads per workgroup in all FFT kernels

// In-LDS Stockham FFT of BATCH sequences of N = 2^LOG2N points.
//   ROWS = true : sequence b is contiguous:      element (b, i) at b*N + i
//   ROWS = false: sequences are interleaved:     element (b, i) at i*BATCH + b   (column tiles)
// tw[m] = exp(-2 pi i m / N).  INV = true computes the unnormalised inverse.  All threads of the
// workgroup must call it; it begins and ends with the data visible to every thread.
template <int LOG2N, int BATCH, bool ROWS, bool INV>
__device__ __forceinline__ void fft_lds(cplx *data, const cplx *tw, int tid)
{
	constexpr int N = 1 << LOG2N;
	auto at = [](int b, int i) { return ROWS ? b * N + i : i * BATCH + b; };
	int Ns = 1;
	// ---- radix-4 passes ----
	constexpr int N4 = N / 4;
	constexpr int TOT4 = (N4 > 0 ? N4 : 1) * BATCH;
	constexpr int BPT4 = (TOT4 + NT - 1) / NT;
#pragma unroll 1
	for (int p = 0; p + 2 <= LOG2N; p += 2) {
		cplx v[BPT4][4];
#pragma unroll
		for (int q = 0; q < BPT4; ++q) {
			const int e = tid + q * NT;
			if (e < TOT4) {
				const int j = ROWS ? e % N4 : e / BATCH, b = ROWS ? e / N4 : e % BATCH;
#pragma unroll
				for (int r = 0; r < 4; ++r) v[q][r] = data[at(b, j + r * N4)];
			}
		}
		__syncthreads();
		const int step = N / (4 * Ns);
#pragma unroll
		for (int q = 0; q < BPT4; ++q) {
			const int e = tid + q * NT;
			if (e < TOT4) {
				const int j = ROWS ? e % N4 : e / BATCH, b = ROWS ? e / N4 : e % BATCH;
				const int k = j & (Ns - 1);
				cplx a0 = v[q][0], a1, a2, a3;
				if (INV) { a1 = cmulc(v[q][1], tw[k * step]); a2 = cmulc(v[q][2], tw[2 * k * step]); a3 = cmulc(v[q][3], tw[3 * k * step]); }
				else { a1 = cmul(v[q][1], tw[k * step]); a2 = cmul(v[q][2], tw[2 * k * step]); a3 = cmul(v[q][3], tw[3 * k * step]); }
				const cplx s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = csub(a1, a3);
				// forward: -i * d13 ; inverse: +i * d13
				const cplx jd = INV ? make_double2(-d13.y, d13.x) : make_double2(d13.y, -d13.x);
				const int j0 = ((j - k) << 2) + k;
				data[at(b, j0)] = cadd(s02, s13);
				data[at(b, j0 + Ns)] = cadd(d02, jd);
				data[at(b, j0 + 2 * Ns)] = csub(s02, s13);
				data[at(b, j0 + 3 * Ns)] = csub(d02, jd);
			}
		}
		__syncthreads();
		Ns <<= 2;
	}
	// ---- radix-2 tail ----
	if (LOG2N & 1) {
		constexpr int N2_ = N / 2;
		constexpr int TOT2 = N2_ * BATCH;
		constexpr int BPT2 = (TOT2 + NT - 1) / NT;
		cplx v[BPT2][2];
#pragma unroll
		for (int q = 0; q < BPT2; ++q) {
			const int e = tid + q * NT;
			if (e < TOT2) {
				const int j = ROWS ? e % N2_ : e / BATCH, b = ROWS ? e / N2_ : e % BATCH;
				v[q][0] = data[at(b, j)];
				v[q][1] = data[at(b, j + N2_)];
			}
		}
		__syncthreads();
		const int step = N / (2 * Ns);
#pragma unroll
		for (int q = 0; q < BPT2; ++q) {
			const int e = tid + q * NT;
			if (e < TOT2) {
				const int j = ROWS ? e % N2_ : e / BATCH, b = ROWS ? e / N2_ : e % BATCH;
				const int k = j & (Ns - 1);
				const cplx a0 = v[q][0];
				const cplx a1 = INV ? cmulc(v[q][1], tw[k * step]) : cmul(v[q][1], tw[k * step]);
				const int j0 = ((j - k) << 1) + k;
				data[at(b, j0)] = cadd(a0, a1);
				data[at(b, j0 + Ns)] = csub(a0, a1);
			}
		}
		__syncthreads();
	}
}

// w_N^m from the two-level table: m = hi * 2^log2_lo + lo
__device__ __forceinline__ cplx big_twiddle(const ConvParams &p, long m)
{
	const cplx a = p.tw_hi[m >> p.log2_lo];
	const cplx b = p.tw_lo[m & ((1L << p.log2_lo) - 1)];
	return cmul(a, b);
}

template <int LOG2N1> struct ColCfg {
	static constexpr int N1 = 1 << LOG2N1;
	static constexpr int TW = (LOG2N1 <= 9) ? 16 : 8;       // column tile width (points of n2)
	static constexpr size_t LDS = ((size_t) N1 * TW + N1) * sizeof(cplx);
};

// K1: z (two planar real rings -> one complex sequence) --FFT over n1--> twiddle --> W[pair][k1][n2]
template <int LOG2N1>
__global__ __launch_bounds__(NT) void conv_col_fwd(ConvParams p)
{
	using Cfg = ColCfg<LOG2N1>;
	constexpr int N1 = Cfg::N1, TW = Cfg::TW;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	cplx *data = reinterpret_cast<cplx *>(smem_raw);
	cplx *tw = data + N1 * TW;
	const int tid = threadIdx.x;
	const long n2_0 = (long) blockIdx.x * TW;
	const long pair = p.pair0 + blockIdx.y;
	for (int i = tid; i < N1; i += NT) tw[i] = p.tw_n1[i];
	const long ra = p.pair_rows[2 * pair], rb = p.pair_rows[2 * pair + 1];
	const double *rowa = (ra >= 0) ? p.ring + ra * p.ring_row_stride : nullptr;
	const double *rowb = (rb >= 0) ? p.ring + rb * p.ring_row_stride : nullptr;
	for (int e = tid; e < N1 * TW; e += NT) {
		const int n1 = e / TW, t = e % TW;
		const long n = (long) n1 * p.N2 + n2_0 + t;
		double re = 0.0, im = 0.0;
		if (n < p.valid) {
			const long ri = (p.win_base + n) & p.ring_mask;
			if (rowa) re = rowa[ri];
			if (rowb) im = rowb[ri];
		}
		data[e] = make_double2(re, im);
	}
	__syncthreads();
	fft_lds<LOG2N1, TW, false, false>(data, tw, tid);
	cplx *W = p.W + (pair - p.pair0) * p.N;
	for (int e = tid; e < N1 * TW; e += NT) {
		const int k1 = e / TW, t = e % TW;
		const long n2 = n2_0 + t;
		const cplx w = big_twiddle(p, (n2 * k1) & (p.N - 1));
		W[(long) k1 * p.N2 + n2] = cmul(data[e], w);
	}
}

// K3: W[pair][k1][n2] --IFFT over k1--> y[n1 N2 + n2]; valid outputs scattered into the interleaved slab.
// One workgroup walks all pairs of its stream for one column tile, so that the 16-byte pieces it writes
// into each 64-byte frame are merged in L2 before they reach HBM.
template <int LOG2N1>
__global__ __launch_bounds__(NT) void conv_col_inv(ConvParams p)
{
	using Cfg = ColCfg<LOG2N1>;
	constexpr int N1 = Cfg::N1, TW = Cfg::TW;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	cplx *data = reinterpret_cast<cplx *>(smem_raw);
	cplx *tw = data + N1 * TW;
	const int tid = threadIdx.x;
	const long n2_0 = (long) blockIdx.x * TW;
	const long s = p.stream0 + blockIdx.y;
	for (int i = tid; i < N1; i += NT) tw[i] = p.tw_n1[i];
	double *out = p.out + ((size_t) s * p.out_stride_frames + p.out_frame0) * p.C;
	for (int q = 0; q < p.pairs_per_stream; ++q) {
		const long pair = s * p.pairs_per_stream + q;
		const cplx *W = p.W + (pair - p.pair0) * p.N;
		__syncthreads();
		for (int e = tid; e < N1 * TW; e += NT) {
			const int k1 = e / TW, t = e % TW;
			data[e] = W[(long) k1 * p.N2 + n2_0 + t];
		}
		__syncthreads();
		fft_lds<LOG2N1, TW, false, true>(data, tw, tid);
		const int cha = p.pair_out_ch[2 * q], chb = p.pair_out_ch[2 * q + 1];
		for (int e = tid; e < N1 * TW; e += NT) {
			const int n1 = e / TW, t = e % TW;
			const long f = (long) n1 * p.N2 + n2_0 + t - p.first_n;
			if (f >= 0 && f < p.out_frames) {
				cplx v = data[e];
				if (p.round_f32) { v.x = (double) (float) v.x; v.y = (double) (float) v.y; }
				if (cha >= 0) out[f * p.C + cha] = v.x;
				if (chb >= 0) out[f * p.C + chb] = v.y;
			}
		}
	}
}

constexpr int ROW_LOG2 = FFT_LOG2_N2;
constexpr int ROW_N = 1 << ROW_LOG2;
constexpr int ROWS_PER_WG = 4;

// K2: per row k1: FFT over n2, multiply by the filter spectrum (already scaled by 1/N), IFFT over k2,
// conjugate twiddle.  mode 1: spectrum only (filter preparation): write scale * FFT to p.Hout.
template <int MODE>
__global__ __launch_bounds__(NT) void conv_row(ConvParams p)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	cplx *data = reinterpret_cast<cplx *>(smem_raw);
	cplx *tw = data + ROWS_PER_WG * ROW_N;
	const int tid = threadIdx.x;
	const long k1_0 = (long) blockIdx.x * ROWS_PER_WG;
	const long pair = p.pair0 + blockIdx.y;
	cplx *W = p.W + (pair - p.pair0) * p.N + k1_0 * ROW_N;
	for (int i = tid; i < ROW_N; i += NT) tw[i] = p.tw_n2[i];
	for (int e = tid; e < ROWS_PER_WG * ROW_N; e += NT) data[e] = W[e];
	__syncthreads();
	fft_lds<ROW_LOG2, ROWS_PER_WG, true, false>(data, tw, tid);
	if (MODE == 1) {
		cplx *H = p.Hout + k1_0 * ROW_N;
		for (int e = tid; e < ROWS_PER_WG * ROW_N; e += NT)
			H[e] = make_double2(data[e].x * p.h_scale, data[e].y * p.h_scale);
		return;
	}
	const cplx *H = p.H + p.pair_h[pair] * p.N + k1_0 * ROW_N;
	for (int e = tid; e < ROWS_PER_WG * ROW_N; e += NT) data[e] = cmul(data[e], H[e]);
	__syncthreads();
	fft_lds<ROW_LOG2, ROWS_PER_WG, true, true>(data, tw, tid);
	for (int e = tid; e < ROWS_PER_WG * ROW_N; e += NT) {
		const long k1 = k1_0 + e / ROW_N, n2 = e % ROW_N;
		const cplx w = big_twiddle(p, (n2 * k1) & (p.N - 1));
		W[e] = cmulc(data[e], w);
	}
}

// interleaved slab -> planar rings for the selected channels (+ pass-through of the others to `out`)
__global__ __launch_bounds__(NT) void conv_deinterleave(DeintParams p)
{
	const int s = blockIdx.y;
	const double *in = p.in + (size_t) s * p.in_stride_frames * p.C;
	double *out = p.out ? p.out + (size_t) s * p.out_stride_frames * p.C : nullptr;
	const long n = p.frames * p.C;
	for (long e = (long) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long) gridDim.x * blockDim.x) {
		const long t = e / p.C;
		const int c = (int) (e - t * p.C);
		const int r = p.row_of_channel[c];
		double v = in[e];
		if (r >= 0) {
			if (p.round_f32) v = (double) (float) v;
			p.ring[((size_t) s * p.rows_per_stream + r) * p.ring_row_stride + ((p.pos + t) & p.ring_mask)] = v;
		}
		else if (out) out[e] = v;
	}
}

// direct-form FIR for <= 32 taps (fir.c:43-62, fir_p.c:131-148), bit-exact: the reference scatter-adds each input
// into a circular accumulator in time order, i.e. every output is ((0 + x[j-T+1] h[T-1]) + ... ) + x[j] h[0]
// with separately rounded products and sums.
__global__ __launch_bounds__(NT) void fir_direct_kernel(FirDirectParams p)
{
	const int s = blockIdx.y;
	const double *in = p.in + (size_t) s * p.in_stride_frames * p.C;
	double *out = p.out + (size_t) s * p.out_stride_frames * p.C;
	const double *hr = p.hist_rd + (size_t) s * p.C * FIR_DIRECT_MAX;
	double *hw = p.hist_wr + (size_t) s * p.C * FIR_DIRECT_MAX;
	const long n = p.frames * p.C;
	for (long e = (long) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long) gridDim.x * blockDim.x) {
		const long t = e / p.C;
		const int c = (int) (e - t * p.C);
		const int fc = p.filter_of_channel[c];
		if (fc < 0) { out[e] = in[e]; continue; }
		const double *h = p.taps + (size_t) fc * FIR_DIRECT_MAX;
		double acc = 0.0;
		for (int m = p.T - 1; m >= 0; --m) {
			const long ti = t - m;
			// history slot q holds x[-(q+1)] relative to this block's first frame
			const double x = (ti >= 0) ? in[ti * p.C + c] : hr[c * FIR_DIRECT_MAX + (-ti - 1)];
			acc = __dadd_rn(acc, __dmul_rn(x, h[m]));
		}
		out[e] = acc;
		// new history: the last T-1 inputs of [old history | this block]
		const long back = p.frames - 1 - t;   // 0 for the newest frame
		if (back < p.T - 1) hw[c * FIR_DIRECT_MAX + back] = in[e];
	}
	// when the block is shorter than the history, older entries shift down
	if (p.frames < p.T - 1) {
		const long m = (long) (p.T - 1 - p.frames) * p.C;
		for (long e = (long) blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (long) gridDim.x * blockDim.x) {
			const long q = e / p.C;
			const int c = (int) (e - q * p.C);
			if (p.filter_of_channel[c] >= 0)
				hw[c * FIR_DIRECT_MAX + p.frames + q] = hr[c * FIR_DIRECT_MAX + q];
		}
	}
}

// ------------------------------------------------------------------ launchers

template <int L> static void launch_col(const ConvParams &p, bool inverse, int grid_y, hipStream_t st)
{
	using Cfg = ColCfg<L>;
	static bool attr_set[2] = { false, false };
	const void *fn = inverse ? (const void *) conv_col_inv<L> : (const void *) conv_col_fwd<L>;
	if (!attr_set[inverse]) {
		(void) hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) Cfg::LDS);
		attr_set[inverse] = true;
	}
	dim3 grid((unsigned) (p.N2 / Cfg::TW), grid_y), block(NT);
	if (inverse) hipLaunchKernelGGL(conv_col_inv<L>, grid, block, Cfg::LDS, st, p);
	else hipLaunchKernelGGL(conv_col_fwd<L>, grid, block, Cfg::LDS, st, p);
}

void launch_conv_col(const ConvParams &p, bool inverse, int grid_y, hipStream_t st)
{
	switch (p.log2N1) {
	case 3: launch_col<3>(p, inverse, grid_y, st); break;
	case 4: launch_col<4>(p, inverse, grid_y, st); break;
	case 5: launch_col<5>(p, inverse, grid_y, st); break;
	case 6: launch_col<6>(p, inverse, grid_y, st); break;
	case 7: launch_col<7>(p, inverse, grid_y, st); break;
	case 8: launch_col<8>(p, inverse, grid_y, st); break;
	case 9: launch_col<9>(p, inverse, grid_y, st); break;
	case 10: launch_col<10>(p, inverse, grid_y, st); break;
	default: break;
	}
}

void launch_conv_row(const ConvParams &p, int mode, int n_pairs, hipStream_t st)
{
	const size_t lds = ((size_t) ROWS_PER_WG * ROW_N + ROW_N) * sizeof(cplx);
	dim3 grid((unsigned) (p.N1 / ROWS_PER_WG), n_pairs), block(NT);
	if (mode == 1) hipLaunchKernelGGL(conv_row<1>, grid, block, lds, st, p);
	else hipLaunchKernelGGL(conv_row<0>, grid, block, lds, st, p);
}

void launch_deinterleave(const DeintParams &p, int n_streams, hipStream_t st)
{
	const long n = p.frames * p.C;
	if (n <= 0) return;
	long blocks = (n + NT - 1) / NT;
	if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(conv_deinterleave, dim3((unsigned) blocks, n_streams), dim3(NT), 0, st, p);
}

void launch_fir_direct(const FirDirectParams &p, int n_streams, hipStream_t st)
{
	long n = p.frames * p.C;
	if (n <= 0) return;
	long blocks = (n + NT - 1) / NT;
	if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(fir_direct_kernel, dim3((unsigned) blocks, n_streams), dim3(NT), 0, st, p);
}

}  // namespace dspamd
