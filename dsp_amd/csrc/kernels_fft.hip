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
//    multiply) so that no multi-megabyte table is streamed.
//  * LDS-resident Stockham passes, radix 8 in registers (radix 4 / 2 for the remainder, done first where its
//    twiddles are trivial), 64-wide waves, 16-byte ds accesses; row tiles are padded by one point per 16 so that
//    the strided Stockham stores are bank-conflict free; column tiles are 16 points wide so every global access
//    is a 256 B run; global loads are issued in register batches ahead of the LDS stores.
#include <hip/hip_runtime.h>
#include "kparams.h"
#include "fft_params.h"

namespace dspamd {

typedef double2 cplx;

__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cmulc(cplx a, cplx b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward) or +i (inverse)
template <bool INV> __device__ __forceinline__ cplx mul_mi(cplx a) { return INV ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x); }

constexpr int NT = 256;   // threads per workgroup in all FFT kernels

// ---- in-register DFTs, natural-order output ----
template <bool INV> __device__ __forceinline__ void dft2(cplx &a, cplx &b)
{
	const cplx s = cadd(a, b), d = csub(a, b);
	a = s; b = d;
}

template <bool INV> __device__ __forceinline__ void dft4(cplx &c0, cplx &c1, cplx &c2, cplx &c3)
{
	const cplx s02 = cadd(c0, c2), d02 = csub(c0, c2), s13 = cadd(c1, c3), d13 = mul_mi<INV>(csub(c1, c3));
	c0 = cadd(s02, s13); c1 = cadd(d02, d13); c2 = csub(s02, s13); c3 = csub(d02, d13);
}

template <bool INV> __device__ __forceinline__ void dft8(cplx (&v)[8])
{
	constexpr double h = 0.70710678118654752440;
	cplx a0 = cadd(v[0], v[4]), a1 = cadd(v[1], v[5]), a2 = cadd(v[2], v[6]), a3 = cadd(v[3], v[7]);
	cplx b0 = csub(v[0], v[4]), b1 = csub(v[1], v[5]), b2 = csub(v[2], v[6]), b3 = csub(v[3], v[7]);
	// b_i *= w8^i,  w8 = exp(-+ i pi/4)
	b1 = INV ? make_double2((b1.x - b1.y) * h, (b1.x + b1.y) * h) : make_double2((b1.x + b1.y) * h, (b1.y - b1.x) * h);
	b2 = mul_mi<INV>(b2);
	b3 = INV ? make_double2(-(b3.x + b3.y) * h, (b3.x - b3.y) * h) : make_double2((b3.y - b3.x) * h, -(b3.x + b3.y) * h);
	dft4<INV>(a0, a1, a2, a3);
	dft4<INV>(b0, b1, b2, b3);
	v[0] = a0; v[1] = b0; v[2] = a1; v[3] = b1; v[4] = a2; v[5] = b2; v[6] = a3; v[7] = b3;
}

template <int R, bool INV> __device__ __forceinline__ void dftR(cplx (&v)[R])
{
	if constexpr (R == 8) dft8<INV>(v);
	else if constexpr (R == 4) dft4<INV>(v[0], v[1], v[2], v[3]);
	else dft2<INV>(v[0], v[1]);
}

// LDS addressing.  ROWS: sequence b contiguous, one pad point per 16 (bank-conflict-free Stockham stores).
// !ROWS: sequences interleaved, element (b, i) at i*BATCH + b (column tiles; BATCH*16 B contiguous per i).
template <int LOG2N, int BATCH, bool ROWS> struct LdsMap {
	static constexpr int N = 1 << LOG2N;
	static constexpr int ROW_PITCH = N + (N >> 4);
	static constexpr int SIZE = ROWS ? BATCH * ROW_PITCH : BATCH * N;
	__device__ static __forceinline__ int at(int b, int i) { return ROWS ? b * ROW_PITCH + i + (i >> 4) : i * BATCH + b; }
};

// one Stockham pass of radix R with accumulated stride Ns
template <int LOG2N, int BATCH, bool ROWS, bool INV, int R>
__device__ __forceinline__ void fft_pass(cplx *data, const cplx *tw, int tid, int Ns)
{
	using M = LdsMap<LOG2N, BATCH, ROWS>;
	constexpr int N = 1 << LOG2N;
	constexpr int NR = N / R;
	constexpr int TOT = NR * BATCH;
	constexpr int BPT = (TOT + NT - 1) / NT;
	cplx v[BPT][R];
#pragma unroll
	for (int q = 0; q < BPT; ++q) {
		const int e = tid + q * NT;
		if (e < TOT) {
			const int j = ROWS ? e % NR : e / BATCH, b = ROWS ? e / NR : e % BATCH;
#pragma unroll
			for (int r = 0; r < R; ++r) v[q][r] = data[M::at(b, j + r * NR)];
		}
	}
	__syncthreads();
	const int step = N / (R * Ns);
#pragma unroll
	for (int q = 0; q < BPT; ++q) {
		const int e = tid + q * NT;
		if (e < TOT) {
			const int j = ROWS ? e % NR : e / BATCH, b = ROWS ? e / NR : e % BATCH;
			const int k = j & (Ns - 1);
			if (Ns > 1) {
#pragma unroll
				for (int r = 1; r < R; ++r) {
					const cplx w = tw[r * k * step];
					v[q][r] = INV ? cmulc(v[q][r], w) : cmul(v[q][r], w);
				}
			}
			dftR<R, INV>(v[q]);
			const int j0 = (j - k) * R + k;
#pragma unroll
			for (int r = 0; r < R; ++r) data[M::at(b, j0 + r * Ns)] = v[q][r];
		}
	}
	__syncthreads();
}

// In-LDS Stockham FFT of BATCH sequences of N = 2^LOG2N points; tw[m] = exp(-2 pi i m / N).
// INV computes the unnormalised inverse.  All threads of the workgroup must call it; the data must be
// visible to every thread on entry (barrier by the caller) and is visible on exit.
template <int LOG2N, int BATCH, bool ROWS, bool INV>
__device__ __forceinline__ void fft_lds(cplx *data, const cplx *tw, int tid)
{
	int Ns = 1;
	if constexpr (LOG2N % 3 == 1) { fft_pass<LOG2N, BATCH, ROWS, INV, 2>(data, tw, tid, Ns); Ns *= 2; }
	if constexpr (LOG2N % 3 == 2) { fft_pass<LOG2N, BATCH, ROWS, INV, 4>(data, tw, tid, Ns); Ns *= 4; }
#pragma unroll 1
	for (int p = 0; p < LOG2N / 3; ++p) { fft_pass<LOG2N, BATCH, ROWS, INV, 8>(data, tw, tid, Ns); Ns *= 8; }
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
	static constexpr int TW = 16;                                     // column tile width (points of n2): 256 B runs
	static constexpr int EPT = (N1 * TW) / NT > 0 ? (N1 * TW) / NT : 1;   // points per thread
	static constexpr size_t LDS = ((size_t) N1 * TW + N1) * sizeof(cplx);
};

// K1: z (two planar real rings -> one complex sequence) --FFT over n1--> twiddle --> W[pair][k1][n2]
template <int LOG2N1>
__global__ __launch_bounds__(NT) void conv_col_fwd(ConvParams p)
{
	using Cfg = ColCfg<LOG2N1>;
	constexpr int N1 = Cfg::N1, TW = Cfg::TW, EPT = Cfg::EPT;
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
	constexpr int BATCHL = EPT < 8 ? EPT : 8;
#pragma unroll 1
	for (int q0 = 0; q0 < EPT; q0 += BATCHL) {
		double re[BATCHL], im[BATCHL];
#pragma unroll
		for (int q = 0; q < BATCHL; ++q) {   // issue all loads of the batch before any LDS store
			const int e = tid + (q0 + q) * NT;
			const int n1 = e / TW, t = e % TW;
			const long n = (long) n1 * p.N2 + n2_0 + t;
			re[q] = 0.0; im[q] = 0.0;
			if (e < N1 * TW && n < p.valid) {
				const long ri = (p.win_base + n) & p.ring_mask;
				if (rowa) re[q] = rowa[ri];
				if (rowb) im[q] = rowb[ri];
			}
		}
#pragma unroll
		for (int q = 0; q < BATCHL; ++q) {
			const int e = tid + (q0 + q) * NT;
			if (e < N1 * TW) data[e] = make_double2(re[q], im[q]);
		}
	}
	__syncthreads();
	fft_lds<LOG2N1, TW, false, false>(data, tw, tid);
	cplx *W = p.W + (pair - p.pair0) * p.N;
#pragma unroll 4
	for (int e = tid; e < N1 * TW; e += NT) {
		const int k1 = e / TW, t = e % TW;
		const long n2 = n2_0 + t;
		const cplx w = big_twiddle(p, (n2 * k1) & (p.N - 1));
		W[(long) k1 * p.N2 + n2] = cmul(data[e], w);
	}
}

// K3: W[pair][k1][n2] --IFFT over k1--> y[n1 N2 + n2]; valid outputs scattered into the interleaved slab.
// blockIdx.x enumerates (tile, stream, pair-in-stream) so that the workgroups writing the 16-byte pieces of the
// same 64-byte frames are dispatched back to back on the SAME XCD (block b runs on XCD b % 8): their partial
// writes then meet in that XCD's L2 instead of reaching HBM as separate 32-byte sectors.
template <int LOG2N1>
__global__ __launch_bounds__(NT) void conv_col_inv(ConvParams p)
{
	using Cfg = ColCfg<LOG2N1>;
	constexpr int N1 = Cfg::N1, TW = Cfg::TW, EPT = Cfg::EPT;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	cplx *data = reinterpret_cast<cplx *>(smem_raw);
	cplx *tw = data + N1 * TW;
	const int tid = threadIdx.x;
	// decode: linear id L = (g_hi * pps + q) * 8 + x, group g = g_hi * 8 + x, group -> (stream-in-chunk, tile)
	const long L = blockIdx.x;
	const int x = (int) (L & 7);
	const long r = L >> 3;
	const int q = (int) (r % p.pairs_per_stream);
	const long g = (r / p.pairs_per_stream) * 8 + x;
	const long n_tiles = p.N2 / TW;
	if (g >= n_tiles * p.n_streams_launch) return;
	const long tile = g % n_tiles;
	const long s = p.stream0 + g / n_tiles;
	const long n2_0 = tile * TW;
	for (int i = tid; i < N1; i += NT) tw[i] = p.tw_n1[i];
	const long pair = s * p.pairs_per_stream + q;
	const cplx *W = p.W + (pair - p.pair0) * p.N;
	constexpr int BATCHL = EPT < 8 ? EPT : 8;
#pragma unroll 1
	for (int q0 = 0; q0 < EPT; q0 += BATCHL) {
		cplx v[BATCHL];
#pragma unroll
		for (int qq = 0; qq < BATCHL; ++qq) {
			const int e = tid + (q0 + qq) * NT;
			if (e < N1 * TW) v[qq] = W[(long) (e / TW) * p.N2 + n2_0 + (e % TW)];
		}
#pragma unroll
		for (int qq = 0; qq < BATCHL; ++qq) {
			const int e = tid + (q0 + qq) * NT;
			if (e < N1 * TW) data[e] = v[qq];
		}
	}
	__syncthreads();
	fft_lds<LOG2N1, TW, false, true>(data, tw, tid);
	double *out = p.out + ((size_t) s * p.out_stride_frames + p.out_frame0) * p.C;
	const int cha = p.pair_out_ch[2 * q], chb = p.pair_out_ch[2 * q + 1];
	const bool wide = (chb == cha + 1) && ((cha & 1) == 0) && ((p.C & 1) == 0) && ((((size_t) out) & 15) == 0);
#pragma unroll 4
	for (int e = tid; e < N1 * TW; e += NT) {
		const int n1 = e / TW, t = e % TW;
		const long f = (long) n1 * p.N2 + n2_0 + t - p.first_n;
		if (f >= 0 && f < p.out_frames) {
			cplx v = data[e];
			if (p.round_f32) { v.x = (double) (float) v.x; v.y = (double) (float) v.y; }
			if (wide) *reinterpret_cast<cplx *>(out + f * p.C + cha) = v;
			else {
				if (cha >= 0) out[f * p.C + cha] = v.x;
				if (chb >= 0) out[f * p.C + chb] = v.y;
			}
		}
	}
}

// rows per workgroup so that a workgroup holds ~2048 points
template <int LOG2N2> struct RowCfg {
	static constexpr int N2 = 1 << LOG2N2;
	static constexpr int RPW = (2048 / N2) > 0 ? 2048 / N2 : 1;
	using M = LdsMap<LOG2N2, RPW, true>;
	static constexpr size_t LDS = ((size_t) M::SIZE + N2) * sizeof(cplx);
	static constexpr int EPT = RPW * N2 / NT;
};

// K2: per row k1: FFT over n2, multiply by the filter spectrum (already scaled by 1/N), IFFT over k2,
// conjugate twiddle.  MODE 1: spectrum only (filter preparation): write scale * FFT to p.Hout.
template <int LOG2N2, int MODE>
__global__ __launch_bounds__(NT) void conv_row(ConvParams p)
{
	using Cfg = RowCfg<LOG2N2>;
	using M = typename Cfg::M;
	constexpr int N2 = Cfg::N2, RPW = Cfg::RPW, EPT = Cfg::EPT;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	cplx *data = reinterpret_cast<cplx *>(smem_raw);
	cplx *tw = data + M::SIZE;
	const int tid = threadIdx.x;
	const long k1_0 = (long) blockIdx.x * RPW;
	const long pair = p.pair0 + blockIdx.y;
	cplx *W = p.W + (pair - p.pair0) * p.N + k1_0 * N2;
	for (int i = tid; i < N2; i += NT) tw[i] = p.tw_n2[i];
	{
		cplx v[EPT];
#pragma unroll
		for (int q = 0; q < EPT; ++q) v[q] = W[tid + q * NT];
#pragma unroll
		for (int q = 0; q < EPT; ++q) { const int e = tid + q * NT; data[M::at(e / N2, e % N2)] = v[q]; }
	}
	__syncthreads();
	fft_lds<LOG2N2, RPW, true, false>(data, tw, tid);
	if (MODE == 1) {
		cplx *H = p.Hout + k1_0 * N2;
#pragma unroll
		for (int q = 0; q < EPT; ++q) {
			const int e = tid + q * NT;
			const cplx d = data[M::at(e / N2, e % N2)];
			H[e] = make_double2(d.x * p.h_scale, d.y * p.h_scale);
		}
		return;
	}
	const cplx *H = p.H + p.pair_h[pair] * p.N + k1_0 * N2;
	{
		cplx h[EPT];
#pragma unroll
		for (int q = 0; q < EPT; ++q) h[q] = H[tid + q * NT];
#pragma unroll
		for (int q = 0; q < EPT; ++q) {
			const int e = tid + q * NT;
			const int a = M::at(e / N2, e % N2);
			data[a] = cmul(data[a], h[q]);
		}
	}
	__syncthreads();
	fft_lds<LOG2N2, RPW, true, true>(data, tw, tid);
#pragma unroll
	for (int q = 0; q < EPT; ++q) {
		const int e = tid + q * NT;
		const long k1 = k1_0 + e / N2, n2 = e % N2;
		const cplx w = big_twiddle(p, (n2 * k1) & (p.N - 1));
		W[e] = cmulc(data[M::at(e / N2, e % N2)], w);
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

template <int L> static void launch_col(const ConvParams &p, bool inverse, int n_pairs, hipStream_t st)
{
	using Cfg = ColCfg<L>;
	static bool attr_set[2] = { false, false };
	const void *fn = inverse ? (const void *) conv_col_inv<L> : (const void *) conv_col_fwd<L>;
	if (!attr_set[inverse]) {
		(void) hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) Cfg::LDS);
		attr_set[inverse] = true;
	}
	if (inverse) {
		const long groups = (p.N2 / Cfg::TW) * p.n_streams_launch;
		const long blocks = ((groups + 7) / 8) * 8 * p.pairs_per_stream;
		hipLaunchKernelGGL(conv_col_inv<L>, dim3((unsigned) blocks), dim3(NT), Cfg::LDS, st, p);
	}
	else hipLaunchKernelGGL(conv_col_fwd<L>, dim3((unsigned) (p.N2 / Cfg::TW), n_pairs), dim3(NT), Cfg::LDS, st, p);
}

void launch_conv_col(const ConvParams &p, bool inverse, int n_pairs, hipStream_t st)
{
	switch (p.log2N1) {
	case 3: launch_col<3>(p, inverse, n_pairs, st); break;
	case 4: launch_col<4>(p, inverse, n_pairs, st); break;
	case 5: launch_col<5>(p, inverse, n_pairs, st); break;
	case 6: launch_col<6>(p, inverse, n_pairs, st); break;
	case 7: launch_col<7>(p, inverse, n_pairs, st); break;
	case 8: launch_col<8>(p, inverse, n_pairs, st); break;
	default: break;
	}
}

template <int L2> static void launch_row(const ConvParams &p, int mode, int n_pairs, hipStream_t st)
{
	using Cfg = RowCfg<L2>;
	static bool attr_set[2] = { false, false };
	if (!attr_set[mode]) {
		const void *fn = mode ? (const void *) conv_row<L2, 1> : (const void *) conv_row<L2, 0>;
		(void) hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) Cfg::LDS);
		attr_set[mode] = true;
	}
	dim3 grid((unsigned) (p.N1 / Cfg::RPW), n_pairs), block(NT);
	if (mode == 1) hipLaunchKernelGGL((conv_row<L2, 1>), grid, block, Cfg::LDS, st, p);
	else hipLaunchKernelGGL((conv_row<L2, 0>), grid, block, Cfg::LDS, st, p);
}

void launch_conv_row(const ConvParams &p, int mode, int n_pairs, hipStream_t st)
{
	switch (p.log2N2) {
	case 9: launch_row<9>(p, mode, n_pairs, st); break;
	case 10: launch_row<10>(p, mode, n_pairs, st); break;
	case 11: launch_row<11>(p, mode, n_pairs, st); break;
	default: break;
	}
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
