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
//    precision on the host; large twiddles are the product of two table entries (one complex multiply),
//    so no multi-megabyte table is streamed and no power recurrences lose bits.
//  * Register-resident Stockham: every thread owns 16 points of a sequence (positions j + P m, P = N/16) and every
//    pass is radix 16 (or 16/R butterflies of radix R for the last factor), so
//      - the first pass runs on the registers the global loads landed in and the last pass feeds the global
//        stores (and, in K2, the H multiply and the first inverse pass) without touching LDS;
//      - LDS only carries the exchanges BETWEEN passes: one round trip for a 256-point column FFT, two for a
//        row FFT of up to 4096 points (a 64-wide wave owns a whole 1024-point row: no workgroup barriers);
//      - every global access is a 16-byte lane access in runs of >= 128 B; K3 gathers the pairs of a stream in
//        one workgroup so that whole 64-byte frames leave in 512-byte runs.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdint>
#include "kparams.h"
#include "fft_params.h"
#include "pcm_device.h"

namespace dspamd {

typedef double2 cplx;

__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cmulc(cplx a, cplx b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward) or +i (inverse)
template <bool INV> __device__ __forceinline__ cplx mul_mi(cplx a) { return INV ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x); }
// multiply by the constant (wr - i wi) (forward) or its conjugate (inverse)
template <bool INV> __device__ __forceinline__ cplx mul_w(cplx a, double wr, double wi)
{
	return INV ? make_double2(a.x * wr - a.y * wi, a.y * wr + a.x * wi) : make_double2(a.x * wr + a.y * wi, a.y * wr - a.x * wi);
}

constexpr int NT = 256;   // threads per workgroup (conv_col_inv with 4 pairs per workgroup uses 2 * NT)

// ---- in-register DFTs, natural-order output ----
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

// 16 = 4 x 4: input index n = n1 + 4 n2, output index k = k2 + 4 k1;
//   A[n1][k2] = sum_n2 x[n1 + 4 n2] w4^(n2 k2),   X[k2 + 4 k1] = sum_n1 (A[n1][k2] w16^(n1 k2)) w4^(n1 k1)
template <bool INV> __device__ __forceinline__ void dft16(cplx (&u)[16])
{
	constexpr double c = 0.92387953251128675613, s = 0.38268343236508977173, h = 0.70710678118654752440;
	cplx a[4][4];
#pragma unroll
	for (int n1 = 0; n1 < 4; ++n1) {
		cplx t0 = u[n1], t1 = u[n1 + 4], t2 = u[n1 + 8], t3 = u[n1 + 12];
		dft4<INV>(t0, t1, t2, t3);
		a[n1][0] = t0; a[n1][1] = t1; a[n1][2] = t2; a[n1][3] = t3;
	}
	// w16^1 = (c, -s)  w16^2 = (h, -h)  w16^3 = (s, -c)  w16^4 = -i  w16^6 = (-h, -h)  w16^9 = (-c, s)
	a[1][1] = mul_w<INV>(a[1][1], c, s);
	a[1][2] = INV ? make_double2((a[1][2].x - a[1][2].y) * h, (a[1][2].x + a[1][2].y) * h) : make_double2((a[1][2].x + a[1][2].y) * h, (a[1][2].y - a[1][2].x) * h);
	a[1][3] = mul_w<INV>(a[1][3], s, c);
	a[2][1] = INV ? make_double2((a[2][1].x - a[2][1].y) * h, (a[2][1].x + a[2][1].y) * h) : make_double2((a[2][1].x + a[2][1].y) * h, (a[2][1].y - a[2][1].x) * h);
	a[2][2] = mul_mi<INV>(a[2][2]);
	a[2][3] = INV ? make_double2(-(a[2][3].x + a[2][3].y) * h, (a[2][3].x - a[2][3].y) * h) : make_double2((a[2][3].y - a[2][3].x) * h, -(a[2][3].x + a[2][3].y) * h);
	a[3][1] = mul_w<INV>(a[3][1], s, c);
	a[3][2] = INV ? make_double2(-(a[3][2].x + a[3][2].y) * h, (a[3][2].x - a[3][2].y) * h) : make_double2((a[3][2].y - a[3][2].x) * h, -(a[3][2].x + a[3][2].y) * h);
	a[3][3] = mul_w<INV>(a[3][3], -c, -s);
#pragma unroll
	for (int k2 = 0; k2 < 4; ++k2) {
		cplx t0 = a[0][k2], t1 = a[1][k2], t2 = a[2][k2], t3 = a[3][k2];
		dft4<INV>(t0, t1, t2, t3);
		u[k2] = t0; u[k2 + 4] = t1; u[k2 + 8] = t2; u[k2 + 12] = t3;
	}
}

template <int R, bool INV> __device__ __forceinline__ void dftR(cplx (&v)[R])
{
	if constexpr (R == 16) dft16<INV>(v);
	else if constexpr (R == 8) dft8<INV>(v);
	else if constexpr (R == 4) dft4<INV>(v[0], v[1], v[2], v[3]);
	else if constexpr (R == 2) { const cplx s = cadd(v[0], v[1]), d = csub(v[0], v[1]); v[0] = s; v[1] = d; }
}

// Twiddle providers (LDS tables).  get<M>(e) = exp(-2 pi i e / M).
struct TwCol {            // sequence length NSEQ <= 256: one table of W_NSEQ; the only twiddled pass has M == NSEQ
	const cplx *t;
	template <int M> __device__ __forceinline__ cplx get(int e) const { return t[e]; }
};
// rows: W_256 direct, W_NSEQ as hi[e >> 6] * lo[e & 63].  The lanes of a pass look these up at e = r k with k = the lane's
// low bits: a stride of r slots, i.e. gcd(r, 16)-way bank conflicts on a plain table (8-way for r = 8).  One slot of padding
// per 16 (twpad) makes every power-of-two stride conflict-free.
__device__ __forceinline__ constexpr int twpad(int e) { return e + (e >> 4); }
template <int NSEQ> struct TwRow {
	const cplx *t256, *lo, *hi;             // t256: [twpad(256)], lo: [twpad(64)], hi: [64]
	template <int M> __device__ __forceinline__ cplx get(int e) const
	{
		if constexpr (M == NSEQ && NSEQ > 256) return cmul(hi[e >> 6], lo[twpad(e & 63)]);
		else return t256[twpad(e * (256 / M))];
	}
};

// One Stockham pass on the 16 register-resident points of a thread.  v[m] <-> position j + P m of the sequence
// (P = N / 16); the pass runs 16 / R butterflies b = j + P q of radix R, whose inputs b + (N / R) r are exactly
// v[q + (16 / R) r] in EVERY pass.  Outputs go to LDS at the Stockham positions, or -- in the last pass, where they
// coincide with the input positions -- stay in v.
template <int LOG2N, int R, int NS, bool INV, bool LAST, class Map, class Tw, class T>
__device__ __forceinline__ void pass16(cplx (&v)[16], int j, T *lds, const Map &map, const Tw &tw)
{
	constexpr int N = 1 << LOG2N, P = N / 16, Q = 16 / R;
#pragma unroll
	for (int q = 0; q < Q; ++q) {
		const int b = j + P * q;
		const int k = b & (NS - 1);
		cplx u[R];
#pragma unroll
		for (int r = 0; r < R; ++r) u[r] = v[q + Q * r];
		if constexpr (NS > 1) {
#pragma unroll
			for (int r = 1; r < R; ++r) {
				const cplx w = tw.template get<R * NS>(r * k);
				u[r] = INV ? cmulc(u[r], w) : cmul(u[r], w);
			}
		}
		dftR<R, INV>(u);
		if constexpr (LAST) {
#pragma unroll
			for (int r = 0; r < R; ++r) v[q + Q * r] = u[r];
		}
		else {
			const int j0 = (b - k) * R + k;
#pragma unroll
			for (int r = 0; r < R; ++r) map.store(lds, j0 + NS * r, u[r]);
		}
	}
}

template <int LOG2N, class Map, class T>
__device__ __forceinline__ void gather16(cplx (&v)[16], int j, const T *lds, const Map &map)
{
	constexpr int P = (1 << LOG2N) / 16;
#pragma unroll
	for (int m = 0; m < 16; ++m) map.load(lds, j + P * m, v[m]);
}

// w_N^m from the two-level table: m = hi * 2^log2_lo + lo
__device__ __forceinline__ cplx big_twiddle(const ConvParams &p, long m)
{
	const cplx a = p.tw_hi[m >> p.log2_lo];
	const cplx b = p.tw_lo[m & ((1L << p.log2_lo) - 1)];
	return cmul(a, b);
}

// The inter-pass twiddle of the four-step transform, w_N^(n2 k1), for the 16 rows k1 = j + P m a thread of a column kernel
// holds of column n2: K1 applies it to its results, K3 its conjugate to what it loads (round 3: it used to sit in K2, where
// 4 complex products per point and 32 LDS reads per thread were 7 % of the time of a kernel that is short of issue slots;
// the column kernels are memory-bound with two thirds of their VALU idle).  Two contiguous table look-ups per thread
// (w_N^(n2 j), w_N^(n2 P) from p.tw_col: the lanes of a column group read 256-byte runs), the rest by products four deep.
template <bool INV, int P, int NV = 1>
__device__ __forceinline__ void col_twiddle(const ConvParams &p, long n2, int j, cplx (&v)[16], cplx (*v2)[16] = nullptr)
{
	const cplx s1 = p.tw_col[(long) P * p.N2 + n2];
	const cplx s2 = cmul(s1, s1), s3 = cmul(s2, s1), s4 = cmul(s2, s2);
	cplx a = p.tw_col[(long) j * p.N2 + n2];
	auto apply = [&](int m, cplx w) {
		v[m] = INV ? cmulc(v[m], w) : cmul(v[m], w);
		if constexpr (NV == 2) (*v2)[m] = INV ? cmulc((*v2)[m], w) : cmul((*v2)[m], w);
	};
#pragma unroll
	for (int g = 0; g < 4; ++g) {
		if (g) a = cmul(a, s4);
		apply(4 * g, a);
		apply(4 * g + 1, cmul(a, s1));
		apply(4 * g + 2, cmul(a, s2));
		apply(4 * g + 3, cmul(a, s3));
	}
}

// ------------------------------------------------------------------ column kernels (K1, K3)
//
// A workgroup owns PPS pairs x TW adjacent columns; thread (q, t, j) owns the points n1 = j + P m of column t of
// pair q.  LDS element (q, pos, t) sits at q * QS + pos * TW + t (QS padded so that the PPS lanes of one frame hit
// different banks).  With SPLIT the real and imaginary parts make separate 8-byte round trips through one
// half-size buffer (4 pairs x 256 x 8 points would not leave room for two workgroups per CU otherwise).
template <int LOG2N1, int PPS> struct ColCfg {
	static constexpr int N1 = 1 << LOG2N1, P = N1 / 16;
	static constexpr int THREADS = (PPS == 4) ? 2 * NT : NT;
	static constexpr int TW = THREADS / (P * PPS);
	static constexpr bool SPLIT = false;   // (a half-size real / imaginary exchange was measured twice: round 1 no occupancy gain at 143 VGPRs; round 2 capped at 128 VGPRs for two workgroups per CU: 6.73 against 6.34 ms, 22 spills and two more barriers)
	static constexpr int QS = N1 * TW + (PPS == 4 ? 4 : PPS == 2 ? 8 : 0);
	static constexpr size_t LDS = (LOG2N1 > 4 ? (size_t) PPS * QS * (SPLIT ? sizeof(double) : sizeof(cplx)) : 0) + (size_t) N1 * sizeof(cplx);
};

template <int TW> struct ColMap {           // full complex elements
	int base;                                // q * QS + t
	__device__ __forceinline__ void store(cplx *lds, int pos, cplx v) const { lds[base + pos * TW] = v; }
	__device__ __forceinline__ void load(const cplx *lds, int pos, cplx &v) const { v = lds[base + pos * TW]; }
};
template <int TW, int PART> struct ColMapHalf {   // one component per round trip
	int base;
	__device__ __forceinline__ void store(double *lds, int pos, cplx v) const { lds[base + pos * TW] = PART ? v.y : v.x; }
	__device__ __forceinline__ void load(const double *lds, int pos, cplx &v) const { if (PART) v.y = lds[base + pos * TW]; else v.x = lds[base + pos * TW]; }
};

// the LDS exchange between the two passes of a column FFT: v (pass-1 outputs, Stockham positions) -> v (pass-2 inputs)
template <int LOG2N1, int PPS, bool INV, class Tw>
__device__ __forceinline__ void col_fft(cplx (&v)[16], int q, int t, int j, unsigned char *smem, const Tw &tw)
{
	using Cfg = ColCfg<LOG2N1, PPS>;
	constexpr int N1 = Cfg::N1, TW = Cfg::TW, R2 = N1 / 16;
	if constexpr (LOG2N1 == 4) {
		pass16<4, 16, 1, INV, true>(v, j, (cplx *) nullptr, ColMap<TW>{ 0 }, tw);
	}
	else if constexpr (!Cfg::SPLIT) {
		cplx *lds = reinterpret_cast<cplx *>(smem);
		const ColMap<TW> map{ q * Cfg::QS + t };
		pass16<LOG2N1, 16, 1, INV, false>(v, j, lds, map, tw);
		lds_barrier();
		gather16<LOG2N1>(v, j, lds, map);
		pass16<LOG2N1, R2, 16, INV, true>(v, j, lds, map, tw);
	}
	else {
		double *lds = reinterpret_cast<double *>(smem);
		const ColMapHalf<TW, 0> map_re{ q * Cfg::QS + t };
		const ColMapHalf<TW, 1> map_im{ q * Cfg::QS + t };
		// pass 1 once, into a scratch copy that both half exchanges read
		cplx u[16];
#pragma unroll
		for (int m = 0; m < 16; ++m) u[m] = v[m];
		pass16<LOG2N1, 16, 1, INV, true>(u, j, lds, map_re, tw);     // LAST: results stay in u (u[r] <-> Stockham position 16 j + r)
		constexpr int P = N1 / 16;
#pragma unroll
		for (int r = 0; r < 16; ++r) lds[map_re.base + (16 * j + r) * TW] = u[r].x;
		lds_barrier();
#pragma unroll
		for (int m = 0; m < 16; ++m) v[m].x = lds[map_re.base + (j + P * m) * TW];
		lds_barrier();
#pragma unroll
		for (int r = 0; r < 16; ++r) lds[map_im.base + (16 * j + r) * TW] = u[r].y;
		lds_barrier();
#pragma unroll
		for (int m = 0; m < 16; ++m) v[m].y = lds[map_im.base + (j + P * m) * TW];
		pass16<LOG2N1, R2, 16, INV, true>(v, j, lds, map_re, tw);
	}
}

// K1: z (the pair's ring row: complex samples) --FFT over n1--> W[pair][k1][n2]
// WIRE: direct mode at the START of a pipeline -- the slab holds samples of p.slab_fmt (s16 / s24 / s32 / float), converted
// as they are read (read_buf_<fmt>, pcm_device.h); the ring keeps fp64 samples as always
template <int LOG2N1, bool WIRE = false>
__global__ __launch_bounds__(NT) void conv_col_fwd(ConvParams p)
{
	using Cfg = ColCfg<LOG2N1, 1>;
	constexpr int N1 = Cfg::N1, TW = Cfg::TW, P = Cfg::P;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	cplx *twt = reinterpret_cast<cplx *>(smem_raw + Cfg::LDS - (size_t) N1 * sizeof(cplx));
	const int tid = threadIdx.x;
	const int t = tid % TW, j = tid / TW;
	const long n2 = (long) blockIdx.x * TW + t;
	const long pair = p.pair0 + blockIdx.y;
	for (int i = tid; i < N1; i += NT) twt[i] = p.tw_n1[i];
	const cplx *src = p.ring + pair * p.ring_row_stride;
	cplx v[16];
	if (p.slab) {
		// direct mode: what lies in the current call's input comes from the interleaved slab (16 bytes = the pair's two
		// channels), older frames from the ring; with slab_store the frames that later windows look back at go into the ring
		// on the way
		cplx *ringw = const_cast<cplx *>(src);
		const long s = pair / p.pairs_per_stream, qs = pair % p.pairs_per_stream;
		const int bs = !WIRE ? 8 : (p.slab_fmt == PCM_S16) ? 2 : 4;
		const WordFormat wf_slab = word_format(p.slab_fmt);
		const char *wslab = reinterpret_cast<const char *>(p.slab) + (((size_t) s * p.slab_stride_frames + p.slab_frame0) * p.C + 2 * qs) * bs;
		const cplx *slab = reinterpret_cast<const cplx *>(wslab);
		const long hp = p.C >> 1;
		const long keep_from = p.slab_store ? ((p.in_count > p.first_n) ? p.in_count : p.first_n) : p.N;
#pragma unroll
		for (int m = 0; m < 16; ++m) {
			const long n = (long) (j + P * m) * p.N2 + n2;
			const long fr = n - p.first_n;                        // slab frame relative to slab_frame0 (folded into `slab`)
			if (n >= p.valid) v[m] = make_double2(0.0, 0.0);
			else if (fr + p.slab_frame0 >= 0) {
				if constexpr (WIRE) {
					const char *e = wslab + fr * p.C * bs;
					if (bs == 4) { const uint2 w = *reinterpret_cast<const uint2 *>(e); v[m] = make_double2(pcm_from_word(w.x, wf_slab), pcm_from_word(w.y, wf_slab)); }
					else { const uint32_t w = *reinterpret_cast<const uint32_t *>(e); v[m] = make_double2(pcm_from_s16(w & 0xffffu), pcm_from_s16(w >> 16)); }
				}
				else v[m] = slab[fr * hp];        // (never non-temporal: the frames of a slab are read by the workgroups of all its pairs)
				if (n >= keep_from) ringw[(p.win_base + n) & p.ring_mask] = v[m];
			}
			else v[m] = ld16(src + ((p.win_base + n) & p.ring_mask), p.nt & 1);
		}
	}
	else {
#pragma unroll
		for (int m = 0; m < 16; ++m) {
			const long n = (long) (j + P * m) * p.N2 + n2;
			v[m] = (n < p.valid) ? ld16(src + ((p.win_base + n) & p.ring_mask), p.nt & 1) : make_double2(0.0, 0.0);
		}
	}
	lds_barrier();   // twiddle table visible (the data loads stay in flight)
	col_fft<LOG2N1, 1, false>(v, 0, t, j, smem_raw, TwCol{ twt });
	col_twiddle<false, P>(p, n2, j, v);
	cplx *W = p.W + (pair - p.pair0) * p.w_stride;
#pragma unroll
	for (int m = 0; m < 16; ++m) st16(W + (long) (j + P * m) * p.N2 + n2, v[m], p.nt & 2);
}

// K3: W[pair][k1][n2] --IFFT over k1--> y[n1 N2 + n2]; valid outputs scattered into the interleaved slab.
// blockIdx.y = stream * groups + group; the workgroup holds PPS pairs of that stream, lanes ordered pair-fastest so
// that the 16-byte (re, im) = (channel 2q, 2q+1) pieces of one frame leave from adjacent lanes.
// MODE 0: plain convolution (one phase, output index = input index: the headline path, no index arithmetic beyond an add);
// MODE 1: any number of phases / up / down; MODE 2: the two interleaved phases of a 2x upsampler -- phase 0 is held in
// registers and frames 2q, 2q+1 leave together.
// At the END of a pipeline run from wire format to wire format (p.sink.on, MODE 0 and 2) the stores go through the sink of
// dsp.c:685-699 (TPDF dither, clip() with its statistics, write_buf_<fmt>; pcm_device.h): a thread's 16 outputs are P N2 frames
// apart, so it reaches its first sample's place in the two dither sequences from the byte tables and the others by multiplying
// with A^(P N2 C).  A run-time branch of the SAME kernel, not an instance of its own: between two instances hipcc's choice of
// fma against mul + add in the transform differed in the last bit of some outputs, and fused or not a call must give the same
// samples (138 VGPRs either way; the headline's K3 measured the same 6.4 ms with the branch in place).
template <int LOG2N1, int PPS, int MODE>
__global__ __launch_bounds__((ColCfg<LOG2N1, PPS>::THREADS)) void conv_col_inv(ConvParams p)
{
	using Cfg = ColCfg<LOG2N1, PPS>;
	constexpr int N1 = Cfg::N1, TW = Cfg::TW, P = Cfg::P, THREADS = Cfg::THREADS;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	cplx *twt = reinterpret_cast<cplx *>(smem_raw + Cfg::LDS - (size_t) N1 * sizeof(cplx));
	const int tid = threadIdx.x;
	const int q = tid % PPS, t = (tid / PPS) % TW, j = tid / (PPS * TW);
	const int groups = (p.pairs_per_stream + PPS - 1) / PPS;
	const long s = p.stream0 + blockIdx.y / groups;
	const int qs = (int) (blockIdx.y % groups) * PPS + q;          // pair within the stream
	const bool active = qs < p.pairs_per_stream;
	const long n2 = (long) blockIdx.x * TW + t;
	for (int i = tid; i < N1; i += THREADS) twt[i] = p.tw_n1[i];
	lds_barrier();
	double *out = p.out + (size_t) s * p.out_stride_frames * p.C;
	const int cha = active ? p.pair_out_ch[2 * qs] : -1, chb = active ? p.pair_out_ch[2 * qs + 1] : -1;
	const bool wide = (chb == cha + 1) && ((cha & 1) == 0) && ((p.C & 1) == 0) && ((((size_t) out) & 15) == 0);
	cplx *rout = p.ring_out ? p.ring_out + (s * p.pairs_per_stream + qs) * p.ring_out_stride : nullptr;
	constexpr bool HOLD2 = (MODE == 2), PLAIN = (MODE == 0);
	const WordFormat wf_sink = word_format(p.sink.fmt);
	if constexpr (HOLD2) {
		cplx v0[16], v[16];
		if (active) {
			const cplx *W = p.W + (s * p.pairs_per_stream + qs - p.pair0) * p.w_stride + n2;
#pragma unroll
			for (int m = 0; m < 16; ++m) v0[m] = ld16(W + (long) (j + P * m) * p.N2, p.nt & 16);
#pragma unroll
			for (int m = 0; m < 16; ++m) v[m] = ld16(W + p.phase_stride + (long) (j + P * m) * p.N2, p.nt & 16);
		}
		else {
#pragma unroll
			for (int m = 0; m < 16; ++m) v0[m] = v[m] = make_double2(0.0, 0.0);
		}
		col_twiddle<true, P, 2>(p, n2, j, v0, &v);
		col_fft<LOG2N1, PPS, true>(v0, q, t, j, smem_raw, TwCol{ twt });
		lds_barrier();
		col_fft<LOG2N1, PPS, true>(v, q, t, j, smem_raw, TwCol{ twt });
		if (p.sink.on) {
			// the 2x upsampler at the END of a pipeline: the sink on the four samples a thread holds per m -- frames mo and mo + 1 of its
			// pair; from one m to the next the position in the dither sequences moves by 2 P N2 C samples
			const long f0 = (long) j * p.N2 + n2 - p.first_n, dmo = (long) P * p.N2;
			const bool dither = p.sink.dither_mult != 0.0;
			const int bs = (p.sink.fmt == PCM_DOUBLE) ? 8 : (p.sink.fmt == PCM_S16) ? 2 : 4;
			char *wout = reinterpret_cast<char *>(p.out) + (size_t) s * p.out_stride_frames * p.C * bs;
			int m_first = 16;
#pragma unroll
			for (int m = 15; m >= 0; --m) { const long f = f0 + m * dmo; if (f >= 0 && f < p.in_count) m_first = m; }
			uint32_t ua0 = 0, ua1 = 0, ub0 = 0, ub1 = 0, j0 = 1, j1 = 1, c0 = 1, c1 = 1;
			if (active && dither && m_first < 16) {
				const long mo = 2 * (p.q_blk + f0 + m_first * dmo) - p.k_origin;          // may lie before the block: signed exponents
				const long na = p.sink.samples_before + mo * p.C + (cha >= 0 ? cha : 0) + 1, nb = p.sink.samples_before + mo * p.C + (chb >= 0 ? chb : 0) + 1;
				ua0 = pm_pow_signed<0>(na); ua1 = pm_pow_signed<1>(na);
				ub0 = pm_pow_signed<0>(nb); ub1 = pm_pow_signed<1>(nb);
				j0 = pm_pow<0>((uint64_t) (2 * dmo * p.C)); j1 = pm_pow<1>((uint64_t) (2 * dmo * p.C));
				c0 = pm_pow<0>((uint64_t) p.C); c1 = pm_pow<1>((uint64_t) p.C);
			}
			double peak = 0.0;
			unsigned long long clipped = 0;
			if (active) {
#pragma unroll
				for (int m = 0; m < 16; ++m) {
					const long f = f0 + m * dmo;
					if (f < 0 || f >= p.in_count) continue;
					const long mo = 2 * (p.q_blk + f) - p.k_origin;
					// (both frames written out by hand: with a loop over the two phases in here the compiler left the loop over m rolled
					// and moved v0 / v to scratch memory -- for the plain path of this kernel too)
					auto emit = [&](long fo, double ya, double yb, uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1) {
						if (fo < 0 || fo >= p.out_count) return;
						ya = sink_sample(ya, dither, a0, a1, p.sink.dither_mult, peak, clipped);
						yb = sink_sample(yb, dither, b0, b1, p.sink.dither_mult, peak, clipped);
						// (whole pairs only -- the host asks for the sink here only when every pair is two adjacent channels of an
						// aligned slab: the element-wise stores of the plain form would make this loop too large to be unrolled)
						char *dst = wout + (fo * p.C + cha) * bs;
						if (bs == 8) *reinterpret_cast<cplx *>(dst) = make_double2(ya, yb);
						else if (bs == 4) *reinterpret_cast<uint2 *>(dst) = make_uint2(pcm_to_word(ya, wf_sink), pcm_to_word(yb, wf_sink));
						else *reinterpret_cast<uint32_t *>(dst) = pcm_to_s16(ya) | (pcm_to_s16(yb) << 16);
					};
					emit(mo, v0[m].x, v0[m].y, ua0, ua1, ub0, ub1);
					// the generator values of frame mo + 1 are C samples on
					emit(mo + 1, v[m].x, v[m].y, dither ? pm_mul(ua0, c0) : 0u, dither ? pm_mul(ua1, c1) : 0u, dither ? pm_mul(ub0, c0) : 0u, dither ? pm_mul(ub1, c1) : 0u);
					if (dither) { ua0 = pm_mul(ua0, j0); ua1 = pm_mul(ua1, j1); ub0 = pm_mul(ub0, j0); ub1 = pm_mul(ub1, j1); }
				}
			}
			if (p.sink.stats) sink_stats_block(p.sink.stats, s, peak, clipped);
			return;
		}
		if (!active) return;
#pragma unroll
		for (int m = 0; m < 16; ++m) {
			const long f = (long) (j + P * m) * p.N2 + n2 - p.first_n;
			if (f < 0 || f >= p.in_count) continue;
			const long mo = 2 * (p.q_blk + f) - p.k_origin;
			if (rout) {
				if (chb < 0) { v0[m].y = 0.0; v[m].y = 0.0; }
				if (mo >= 0 && mo < p.out_count) rout[(p.ring_out_pos + mo) & p.ring_out_mask] = v0[m];
				if (mo + 1 >= 0 && mo + 1 < p.out_count) rout[(p.ring_out_pos + mo + 1) & p.ring_out_mask] = v[m];
			}
			else if (wide) {
				if (mo >= 0 && mo < p.out_count) *reinterpret_cast<cplx *>(out + mo * p.C + cha) = v0[m];
				if (mo + 1 >= 0 && mo + 1 < p.out_count) *reinterpret_cast<cplx *>(out + (mo + 1) * p.C + cha) = v[m];
			}
			else {
				if (mo >= 0 && mo < p.out_count) { if (cha >= 0) out[mo * p.C + cha] = v0[m].x; if (chb >= 0) out[mo * p.C + chb] = v0[m].y; }
				if (mo + 1 >= 0 && mo + 1 < p.out_count) { if (cha >= 0) out[(mo + 1) * p.C + cha] = v[m].x; if (chb >= 0) out[(mo + 1) * p.C + chb] = v[m].y; }
			}
		}
		return;
	}
	const int nph = PLAIN ? 1 : p.nph;
	for (int ph = 0; ph < nph; ++ph) {
		cplx v[16];
		if (active) {
			const cplx *W = p.W + (PLAIN ? 0 : ph * p.phase_stride) + (s * p.pairs_per_stream + qs - p.pair0) * p.w_stride + n2;
#pragma unroll
			for (int m = 0; m < 16; ++m) v[m] = ld16(W + (long) (j + P * m) * p.N2, p.nt & 16);
		}
		else {
#pragma unroll
			for (int m = 0; m < 16; ++m) v[m] = make_double2(0.0, 0.0);
		}
		if (ph > 0) lds_barrier();   // the previous phase's exchange has been read by everyone
		col_twiddle<true, P>(p, n2, j, v);
		col_fft<LOG2N1, PPS, true>(v, q, t, j, smem_raw, TwCol{ twt });
		if (PLAIN && p.sink.on) {
			// window sample of v[m]: f(m) = f0 + m P N2 -> output frame q_blk + f - k_origin; the valid m are a contiguous range
			const long f0 = (long) j * p.N2 + n2 - p.first_n, dmo = (long) P * p.N2;
			const long lo = (p.k_origin - p.q_blk > 0) ? p.k_origin - p.q_blk : 0;             // f >= lo  <=>  mo >= 0
			const long hi = (p.in_count < p.out_count + p.k_origin - p.q_blk) ? p.in_count : p.out_count + p.k_origin - p.q_blk;   // f < hi
			const bool dither = p.sink.dither_mult != 0.0;
			const int bs = (p.sink.fmt == PCM_DOUBLE) ? 8 : (p.sink.fmt == PCM_S16) ? 2 : 4;
			char *wout = reinterpret_cast<char *>(p.out) + (size_t) s * p.out_stride_frames * p.C * bs;
			const bool wpair = (chb == cha + 1) && ((cha & 1) == 0) && ((p.C & 1) == 0) && ((((size_t) wout) & 15) == 0);
			int m_first = 16;
#pragma unroll
			for (int m = 15; m >= 0; --m) { const long f = f0 + m * dmo; if (f >= lo && f < hi) m_first = m; }
			uint32_t ua0 = 0, ua1 = 0, ub0 = 0, ub1 = 0, j0 = 1, j1 = 1;
			if (active && dither && m_first < 16) {
				const long mo = p.q_blk + f0 + m_first * dmo - p.k_origin;
				const uint64_t na = (uint64_t) (p.sink.samples_before + mo * p.C + (cha >= 0 ? cha : 0)) + 1;
				const uint64_t nb = (uint64_t) (p.sink.samples_before + mo * p.C + (chb >= 0 ? chb : 0)) + 1;
				ua0 = pm_pow<0>(na); ua1 = pm_pow<1>(na);
				if (chb == cha + 1) { ub0 = pm_mul(ua0, PM_A0); ub1 = pm_mul(ua1, PM_A1); }
				else { ub0 = pm_pow<0>(nb); ub1 = pm_pow<1>(nb); }
				j0 = pm_pow<0>((uint64_t) dmo * p.C); j1 = pm_pow<1>((uint64_t) dmo * p.C);
			}
			double peak = 0.0;
			unsigned long long clipped = 0;
			if (active) {
#pragma unroll
				for (int m = 0; m < 16; ++m) {
					const long f = f0 + m * dmo;
					if (f < lo || f >= hi) continue;
					const long mo = p.q_blk + f - p.k_origin;
					double ya = v[m].x, yb = v[m].y;
					if (p.round_f32) { ya = (double) (float) ya; yb = (double) (float) yb; }
					if (cha >= 0) ya = sink_sample(ya, dither, ua0, ua1, p.sink.dither_mult, peak, clipped);
					if (chb >= 0) yb = sink_sample(yb, dither, ub0, ub1, p.sink.dither_mult, peak, clipped);
					if (dither) { ua0 = pm_mul(ua0, j0); ua1 = pm_mul(ua1, j1); ub0 = pm_mul(ub0, j0); ub1 = pm_mul(ub1, j1); }
					if (wpair) {
						char *dst = wout + (mo * p.C + cha) * bs;
						if (bs == 8) st16(reinterpret_cast<cplx *>(dst), make_double2(ya, yb), p.nt & 32);
						else if (bs == 4) *reinterpret_cast<uint2 *>(dst) = make_uint2(pcm_to_word(ya, wf_sink), pcm_to_word(yb, wf_sink));
						else *reinterpret_cast<uint32_t *>(dst) = pcm_to_s16(ya) | (pcm_to_s16(yb) << 16);
					}
					else {
						if (cha >= 0) pcm_store(wout, p.sink.fmt, mo * p.C + cha, ya);
						if (chb >= 0) pcm_store(wout, p.sink.fmt, mo * p.C + chb, yb);
					}
				}
			}
			if (p.sink.stats) sink_stats_block(p.sink.stats, s, peak, clipped);
			continue;
		}
		if (!active) continue;
#pragma unroll
		for (int m = 0; m < 16; ++m) {
			const long f = (long) (j + P * m) * p.N2 + n2 - p.first_n;
			if (f < 0 || f >= p.in_count) continue;
			long mo;
			if constexpr (PLAIN) mo = p.q_blk + f - p.k_origin;
			else {
				mo = p.up * (p.q_blk + f) + ph;
				if (p.down > 1) { if (mo % p.down) continue; mo /= p.down; }
				mo -= p.k_origin;
			}
			if (mo < 0 || mo >= p.out_count) continue;
			cplx y = v[m];
			if (p.round_f32) { y.x = (double) (float) y.x; y.y = (double) (float) y.y; }
			if (rout) {
				if (p.ring_out_round_f32) { y.x = (double) (float) y.x; y.y = (double) (float) y.y; }
				if (chb < 0) y.y = 0.0;
				rout[(p.ring_out_pos + mo) & p.ring_out_mask] = y;
			}
			else if (wide) st16(reinterpret_cast<cplx *>(out + mo * p.C + cha), y, p.nt & 32);
			else {
				if (cha >= 0) out[mo * p.C + cha] = y.x;
				if (chb >= 0) out[mo * p.C + chb] = y.y;
			}
		}
	}
}

// ------------------------------------------------------------------ row kernel (K2)
//
// Rows of N2 = 16 * 16 * R3 points, P = N2 / 16 threads per row, 4096 / N2 rows per workgroup.  Padded LDS rows
// (one point per 16) keep the strided Stockham stores conflict-free.  A row of <= 1024 points belongs to a single
// wave, which then needs no workgroup barrier between passes.
template <int LOG2N2> struct RowCfg {
	static constexpr int N2 = 1 << LOG2N2, P = N2 / 16, RPW = NT / P, R3 = N2 / 256;
	static constexpr int PITCH = N2 + N2 / 16;
	static constexpr int T256 = 272, TLO = 68;            // padded table lengths (twpad)
	static constexpr int NTW = T256 + TLO + 64;             // W_256, W_N2 lo / hi
	static constexpr size_t LDS = ((size_t) RPW * PITCH + NTW) * sizeof(cplx);
	static constexpr bool WAVE_LOCAL = (P <= 64);
};

// Where point `pos` of a row lives: the low three bits of its 16-byte slot XORed with bits 4..6 of pos.  Conflict-free for
// both access shapes of the exchanges on this LDS (MI355X_MICROARCH.md, LDS table): a Stockham store instruction writes
// 16 j + r or 16 (j - k) + k + 16 r from 8 contiguous lanes (a ds_write_b128 is served in groups of 8 lanes x 4 banks of 32:
// the slots must differ mod 8), a gather reads j + P m from the lane groups {0-3, 12-15, 20-27} ... of a ds_read_b128 (16
// lanes x 4 banks of 64: the slots must differ mod 16).  Round 2's padding (pos + pos / 16) served the stores but left a
// two-way conflict in every gather group -- lanes 12 and 27 -- and SQ_LDS_BANK_CONFLICT at 28 % of the LDS cycles of K2.
struct RowMap {
	int base;                                // row * PITCH
	__device__ __forceinline__ static int slot(int pos) { return pos ^ ((pos >> 4) & 7); }
	__device__ __forceinline__ void store(cplx *lds, int pos, cplx v) const { lds[base + slot(pos)] = v; }
	__device__ __forceinline__ void load(const cplx *lds, int pos, cplx &v) const { v = lds[base + slot(pos)]; }
};

template <bool WAVE_LOCAL> __device__ __forceinline__ void row_sync()
{
	if constexpr (WAVE_LOCAL) {
		// same-wave LDS traffic is processed in order; only the compiler must not move accesses across this point
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
	else lds_barrier();
}

template <int LOG2N2, bool INV, class Tw>
__device__ __forceinline__ void row_fft(cplx (&v)[16], int j, cplx *lds, const RowMap &map, const Tw &tw)
{
	using Cfg = RowCfg<LOG2N2>;
	constexpr bool WL = Cfg::WAVE_LOCAL;
	pass16<LOG2N2, 16, 1, INV, false>(v, j, lds, map, tw);
	row_sync<WL>();
	gather16<LOG2N2>(v, j, lds, map);
	row_sync<WL>();
	pass16<LOG2N2, 16, 16, INV, false>(v, j, lds, map, tw);
	row_sync<WL>();
	gather16<LOG2N2>(v, j, lds, map);
	pass16<LOG2N2, Cfg::R3, 256, INV, true>(v, j, lds, map, tw);
}

// K2: per row k1 (the inter-pass twiddle has been applied by K1): FFT over n2, multiply by the filter spectrum (already scaled
// by 1/N), IFFT over k2 (K3 applies the conjugate twiddle).  MODE 1: spectrum only (filter preparation): write scale * FFT to p.Hout.
template <int LOG2N2, int MODE>
__global__ __launch_bounds__(NT) void conv_row(ConvParams p)
{
	using Cfg = RowCfg<LOG2N2>;
	constexpr int N2 = Cfg::N2, P = Cfg::P, RPW = Cfg::RPW;
	constexpr bool WL = Cfg::WAVE_LOCAL;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	cplx *data = reinterpret_cast<cplx *>(smem_raw);
	cplx *t256 = data + RPW * Cfg::PITCH, *tlo = t256 + Cfg::T256, *thi = tlo + Cfg::TLO;
	const int tid = threadIdx.x;
	const int rw = tid / P, j = tid % P;
	const long k1 = (long) blockIdx.x * RPW + rw;
	const long pair = p.pair0 + blockIdx.y;
	cplx *W = p.W + (pair - p.pair0) * p.w_stride + k1 * N2 + j;
	cplx v[16];
#pragma unroll
	for (int m = 0; m < 16; ++m) v[m] = ld16(W + P * m, p.nt & 4);
	t256[twpad(tid)] = p.tw_n2[tid * (N2 / 256)];
	if (tid < 64) tlo[twpad(tid)] = p.tw_n2[tid];
	else if (tid < 64 + N2 / 64) thi[tid - 64] = p.tw_n2[(tid - 64) * 64];
	lds_barrier();
	const TwRow<N2> tw{ t256, tlo, thi };
	const RowMap map{ rw * Cfg::PITCH };
	row_fft<LOG2N2, false>(v, j, data, map, tw);
	if (MODE == 1) {
		cplx *H = p.Hout + k1 * N2 + j;
#pragma unroll
		for (int m = 0; m < 16; ++m) H[P * m] = make_double2(v[m].x * p.h_scale, v[m].y * p.h_scale);
		return;
	}
	if (MODE == 2) {
		// several filters on the same input (the phases of an integer-ratio resampler): one forward transform, one
		// multiply + inverse transform per phase, each into its own W
		for (int ph = 0; ph < p.nph; ++ph) {
			const cplx *H = p.H + ((long) p.pair_h[pair] * p.nph + ph) * p.N + k1 * N2 + j;
			cplx u[16];
#pragma unroll
			for (int m = 0; m < 16; ++m) u[m] = cmul(v[m], H[P * m]);
			row_sync<WL>();
			row_fft<LOG2N2, true>(u, j, data, map, tw);
			cplx *Wp = W + ph * p.phase_stride;
#pragma unroll
			for (int m = 0; m < 16; ++m) Wp[P * m] = u[m];
		}
		return;
	}
	{
		const cplx *H = p.H + p.pair_h[pair] * p.N + k1 * N2 + j;
		cplx h[16];
#pragma unroll
		for (int m = 0; m < 16; ++m) h[m] = H[P * m];
#pragma unroll
		for (int m = 0; m < 16; ++m) v[m] = cmul(v[m], h[m]);
	}
	row_sync<WL>();      // every forward gather has completed before the inverse passes overwrite the row
	row_fft<LOG2N2, true>(v, j, data, map, tw);
#pragma unroll
	for (int m = 0; m < 16; ++m) st16(W + P * m, v[m], p.nt & 8);
}

// K2, persistent form (plain convolution, one filter shared by every pair): a workgroup keeps ITS rows k1 and walks over the
// pairs.  What the one-shot kernel above redoes per row and pair happens once per workgroup: the filter rows live in
// registers, the pass twiddles in LDS.  The next pair's rows come in by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass) while the current ones are transformed, so the HBM
// stream never waits for a compute phase -- at 244 VGPRs the one-shot kernel holds two workgroups per CU whose load, compute
// and store phases overlap only by chance (2.0 ms for 8.6 GB where the bare access pattern moves them in 1.55,
// scripts/ubench/hbmprobe.hip).  Rows of 2048 / 4096 points take the same path (three radix-16 passes with workgroup
// barriers): the extra barriers hide behind the stream instead of adding to it.
// LDS: landing zone [16][256] points (thread t's point m at m * 256 + t: what every wave fetches is contiguous, as LDS-DMA
// requires, and what a thread reads back is its own wave's) + the padded exchange rows + tables = 139 KB: one workgroup per CU.
// vmcnt runs in order on this chip (loads and stores share it): at the top of an iteration the 16 stores of the previous
// pair are the youngest operations, so `vmcnt(16)` means "this pair's rows have landed".
// Measured at the headline shape (1024-point rows, 8.6 GB per launch): 2.05 -> 1.82 ms.  Without the transforms the same loop
// takes 1.71 ms, without waiting for the landing 1.87: neither the arithmetic nor the read latency is what is left -- a
// second prefetch stage in the accumulation registers (pairs two ahead, 128 KB of reads in flight per CU) changed nothing
// (9.3 against 9.0 ms at 4096-point rows).  5.0 TB/s is what this in-place read-modify-write stream gets from the memory
// system; the bare pattern reads 5.3-5.45 TB/s in scripts/ubench/hbmprobe.hip, a plain copy 6.3.
// NPH = 2: the two polyphase branches of a 2x upsampler (one forward transform, one multiply + inverse transform per
// branch, each into its own W: conv_row<., 2> in persistent form)
template <int LOG2N2, int NPH>
__global__ __launch_bounds__(NT) void conv_row_pipe(ConvParams p, int pairs_per_wg, int n_pairs)
{
	using Cfg = RowCfg<LOG2N2>;
	constexpr int N2 = Cfg::N2, P = Cfg::P, RPW = Cfg::RPW;
	constexpr bool WL = Cfg::WAVE_LOCAL;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	cplx *land = reinterpret_cast<cplx *>(smem_raw);                    // [16][NT]
	cplx *data = land + 16 * NT;                                         // [RPW][PITCH]
	cplx *t256 = data + RPW * Cfg::PITCH, *tlo = t256 + Cfg::T256, *thi = tlo + Cfg::TLO;
	const int tid = threadIdx.x;
	const int rw = tid / P, j = tid % P;
	const long k1 = (long) blockIdx.x * RPW + rw;
	const long q0 = (long) blockIdx.y * pairs_per_wg;
	const long q1 = (q0 + pairs_per_wg < n_pairs) ? q0 + pairs_per_wg : n_pairs;
	if (q0 >= q1) return;
	cplx *W = p.W + k1 * N2 + j;                                         // + pair * N
	// LDS-DMA by hand: through the builtin the compiler books the transfer as an LDS write and parks an `s_waitcnt vmcnt(0)` in
	// front of the next exchange-buffer access -- which would wait for the prefetch it is meant to overlap.  M0 = destination
	// (wave-uniform LDS byte address; the hardware adds lane * 16), saved and restored around the instruction.
	const unsigned land_wave = __builtin_amdgcn_readfirstlane((unsigned) (uintptr_t) (land + (tid & ~63)));
	auto fetch = [&](long q) {
		const cplx *src = W + q * p.w_stride;                                   // (q counts from the launch's first pair, like W)
#pragma unroll
		for (int m = 0; m < 16; ++m) {
			unsigned keep;
			asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
			             : "=&s"(keep) : "v"(src + P * m), "s"(land_wave + (unsigned) (NT * m * sizeof(cplx))) : "memory");
		}
	};
	fetch(q0);
	t256[twpad(tid)] = p.tw_n2[tid * (N2 / 256)];
	if (tid < 64) tlo[twpad(tid)] = p.tw_n2[tid];
	else if (tid < 64 + N2 / 64) thi[tid - 64] = p.tw_n2[(tid - 64) * 64];
	cplx h[NPH][16];
#pragma unroll
	for (int ph = 0; ph < NPH; ++ph) {
		const cplx *H = p.H + (long) ph * p.N + k1 * N2 + j;             // (one shared filter set: pair_h is all zeros)
#pragma unroll
		for (int m = 0; m < 16; ++m) h[ph][m] = H[P * m];
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the first fetch (the compiler does not know about it)
	__syncthreads();                                                     // tables visible
	const TwRow<N2> tw{ t256, tlo, thi };
	const RowMap map{ rw * Cfg::PITCH };
	for (long q = q0; q < q1; ++q) {
		if (q > q0) {                                                    // this pair's rows have landed (see above)
			if (NPH == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
		}
		cplx v[16];
#pragma unroll
		for (int m = 0; m < 16; ++m) v[m] = land[NT * m + tid];
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // own landing slots read: they may be overwritten
		if (q + 1 < q1) fetch(q + 1);
		row_fft<LOG2N2, false>(v, j, data, map, tw);
#pragma unroll
		for (int ph = 0; ph < NPH; ++ph) {
			cplx u[16];
#pragma unroll
			for (int m = 0; m < 16; ++m) u[m] = cmul(v[m], h[ph][m]);
			row_sync<WL>();      // every gather of the transform before has completed before the inverse passes overwrite the row
			row_fft<LOG2N2, true>(u, j, data, map, tw);
			cplx *out = W + (long) ph * p.phase_stride + q * p.w_stride;
#pragma unroll
			for (int m = 0; m < 16; ++m) out[P * m] = u[m];
		}
		if (q + 1 == q1) break;
		row_sync<WL>();      // the last gather of the inverse transform is done before the next forward pass writes the row
	}
}

// K2, persistent form for rows of 2048 / 4096 points (round 3): TWO workgroups per CU instead of a landing zone.
// Rows that span several waves meet at seven workgroup barriers per pair; with the 64 KB landing zone of conv_row_pipe a CU
// holds one workgroup = one wave per SIMD, and butterflies (3.2 us of fp64 issue per 4096-point row pair), LDS exchanges
// (2.2 us) and barriers run strictly one after the other: 8.5 us per row where its 128 KB of HBM traffic need 5.6
// (profiles/r03_clock.json: 34 % VALU-busy).  Here a workgroup keeps only the exchange row and the tables (74 KB) and loads its
// row straight into the registers it transforms; the second workgroup of the CU fills the gaps: while one waits at a barrier
// or for its loads the other issues butterflies.  The filter row stays in registers; the wave's budget is 256 of them (2 waves
// per SIMD), which holds because W is addressed through a buffer descriptor (no per-access address pairs).
template <int LOG2N2>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_row_duo(ConvParams p, int pairs_per_wg, int n_pairs)
{
	using Cfg = RowCfg<LOG2N2>;
	constexpr int N2 = Cfg::N2, P = Cfg::P, RPW = Cfg::RPW;
	constexpr bool WL = Cfg::WAVE_LOCAL;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	cplx *data = reinterpret_cast<cplx *>(smem_raw);                    // [RPW][PITCH]
	cplx *t256 = data + RPW * Cfg::PITCH, *tlo = t256 + Cfg::T256, *thi = tlo + Cfg::TLO;
	const int tid = threadIdx.x;
	const int rw = tid / P, j = tid % P;
	const long k1 = (long) blockIdx.x * RPW + rw;
	const long q0 = (long) blockIdx.y * pairs_per_wg;
	const long q1 = (q0 + pairs_per_wg < n_pairs) ? q0 + pairs_per_wg : n_pairs;
	if (q0 >= q1) return;
	// W of a pair through a buffer descriptor (SGPR base = the pair's W, one per-lane offset register, constant offsets per
	// access): with plain pointers the 16 strided loads and 16 strided stores (4 KB and more apart: beyond the instruction's
	// immediate) each hold a 64-bit address pair -- 64 registers that the two-workgroup budget does not have
	typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
	auto pair_rsrc = [&](long q) { return __builtin_amdgcn_make_buffer_rsrc(p.W + q * p.w_stride, 0, 0x7fffffff, 0x00020000); };
	const int vo = (int) ((k1 * N2 + j) * (long) sizeof(cplx));
	cplx v[16];
	{
		const __amdgpu_buffer_rsrc_t r = pair_rsrc(q0);
#pragma unroll
		for (int m = 0; m < 16; ++m) v[m] = __builtin_bit_cast(cplx, __builtin_amdgcn_raw_buffer_load_b128(r, vo, P * m * (int) sizeof(cplx), 0));
	}
	t256[twpad(tid)] = p.tw_n2[tid * (N2 / 256)];
	if (tid < 64) tlo[twpad(tid)] = p.tw_n2[tid];
	else if (tid < 64 + N2 / 64) thi[tid - 64] = p.tw_n2[(tid - 64) * 64];
	cplx h[16];
	{
		const cplx *H = p.H + k1 * N2 + j;
#pragma unroll
		for (int m = 0; m < 16; ++m) h[m] = H[P * m];
	}
	lds_barrier();                                                       // tables visible
	const TwRow<N2> tw{ t256, tlo, thi };
	const RowMap map{ rw * Cfg::PITCH };
	for (long q = q0; q < q1; ++q) {
		row_fft<LOG2N2, false>(v, j, data, map, tw);
		// (the scheduling fences keep the unrolled element-wise loops from being turned into sixteen loads, then sixteen products ...)
#pragma unroll
		for (int m = 0; m < 16; ++m) { v[m] = cmul(v[m], h[m]); if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0); }
		row_sync<WL>();      // every gather of the forward transform has completed before the inverse passes overwrite the row
		row_fft<LOG2N2, true>(v, j, data, map, tw);
		{
			// (the offset of a 128-bit buffer store rides in the per-lane register, soffset = 0: DESIGN.md section 4.1, the gfx950 hazard)
			const __amdgpu_buffer_rsrc_t r = pair_rsrc(q);
#pragma unroll
			for (int m = 0; m < 16; ++m) {
				__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[m]), r, vo + P * m * (int) sizeof(cplx), 0, 0);
				if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
			}
		}
		if (q + 1 == q1) break;
		{
			const __amdgpu_buffer_rsrc_t r = pair_rsrc(q + 1);
#pragma unroll
			for (int m = 0; m < 16; ++m) v[m] = __builtin_bit_cast(cplx, __builtin_amdgcn_raw_buffer_load_b128(r, vo, P * m * (int) sizeof(cplx), 0));
		}
		row_sync<WL>();      // the last gather of the inverse transform is done before the next forward pass writes the row
	}
}

// K2 for rows of N2 = WV * 1024 points (WV = 2, 4): the row FFT is itself split 4-step style so that all but one
// exchange per direction stay inside a wave.  With n2 = a + 1024 b and k2 = WV ka + kb:
//   forward   Z_kb[a] = w_N2^(a kb) sum_b x[a + 1024 b] w_WV^(b kb)      radix-WV butterflies on registers (a thread
//                                                                        loads its own WV blocks), then ONE
//                                                                        cross-wave exchange: wave kb collects Z_kb
//             X[WV ka + kb] = sum_a Z_kb[a] w_1024^(a ka)                1024-point FFT inside wave kb (no barriers)
//   multiply by H, which filter preparation (MODE 1) stores in this kernel's own (kb, ka) order
//   inverse   the mirror image: 1024-point IFFT inside the wave, conj twiddle, ONE cross-wave exchange, radix-WV
//             butterflies, y[a + 1024 b] leaves in contiguous runs.
// Two workgroup barriers per row instead of the seven of the generic 3-pass kernel.
template <int WV, int MODE>
__global__ __launch_bounds__(NT) void conv_row_big(ConvParams p)
{
	using C10 = RowCfg<10>;
	constexpr int N2 = 1024 * WV, AV = 16 / WV, TR = 64 * WV, ROWS = 4 / WV, PITCH = C10::PITCH;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	cplx *data = reinterpret_cast<cplx *>(smem_raw);                 // [4 sub-rows][PITCH]
	cplx *t256 = data + 4 * PITCH, *tlo = t256 + C10::T256, *thi = tlo + 64, *t1lo = thi + 64, *t1hi = t1lo + C10::TLO;   // t256, t1lo: padded (twpad)
	const int tid = threadIdx.x;
	const int lane = tid & 63, wq = tid >> 6;                        // wq: wave of the workgroup = sub-row buffer
	const int rw = wq / WV, w = wq % WV, tr = w * 64 + lane;
	const long k1 = (long) blockIdx.x * ROWS + rw;
	const long pair = p.pair0 + blockIdx.y;
	cplx *W = p.W + (pair - p.pair0) * p.w_stride + k1 * N2;
	cplx v[16];
#pragma unroll
	for (int i = 0; i < AV; ++i)
#pragma unroll
		for (int b = 0; b < WV; ++b) v[i * WV + b] = W[tr + TR * i + 1024 * b];
	// tables: W_256, W_N2 two-level (cross-wave twiddles), W_1024 two-level (in-wave FFT)
	t256[twpad(tid)] = p.tw_n2[tid * (N2 / 256)];
	if (tid < 64) { tlo[tid] = p.tw_n2[tid]; t1lo[twpad(tid)] = p.tw_n2[tid * WV]; }
	else if (tid < 128) { thi[tid - 64] = (tid - 64 < N2 / 64) ? p.tw_n2[(tid - 64) * 64] : make_double2(0.0, 0.0); }
	else if (tid < 144) t1hi[tid - 128] = p.tw_n2[(tid - 128) * 64 * WV];
	lds_barrier();
	const TwRow<1024> tw1{ t256, t1lo, t1hi };
	const int rbase = rw * WV * PITCH;
	const RowMap mymap{ wq * PITCH };
	// ---- forward: radix-WV over the blocks, twiddle, hand Z_kb to wave kb ----
#pragma unroll
	for (int i = 0; i < AV; ++i) {
		cplx u[WV];
#pragma unroll
		for (int b = 0; b < WV; ++b) u[b] = v[i * WV + b];
		dftR<WV, false>(u);
		const int a = tr + TR * i;
#pragma unroll
		for (int kb = 0; kb < WV; ++kb) {
			if (kb > 0) { const int e = a * kb; u[kb] = cmul(u[kb], cmul(thi[e >> 6], tlo[e & 63])); }
			RowMap{ rbase + kb * PITCH }.store(data, a, u[kb]);
		}
	}
	lds_barrier();
	gather16<10>(v, lane, data, mymap);
	row_sync<true>();
	row_fft<10, false>(v, lane, data, mymap, tw1);
	if (MODE == 1) {
		cplx *H = p.Hout + k1 * N2 + w * 1024 + lane;
#pragma unroll
		for (int m = 0; m < 16; ++m) H[64 * m] = make_double2(v[m].x * p.h_scale, v[m].y * p.h_scale);
		return;
	}
	{
		const cplx *H = p.H + p.pair_h[pair] * p.N + k1 * N2 + w * 1024 + lane;
#pragma unroll
		for (int m = 0; m < 16; ++m) v[m] = cmul(v[m], H[64 * m]);
	}
	// ---- inverse ----
	row_sync<true>();
	row_fft<10, true>(v, lane, data, mymap, tw1);
	row_sync<true>();
#pragma unroll
	for (int m = 0; m < 16; ++m) {
		const int a = lane + 64 * m;
		cplx z = v[m];
		if (w > 0) { const int e = a * w; z = cmulc(z, cmul(thi[e >> 6], tlo[e & 63])); }
		mymap.store(data, a, z);
	}
	lds_barrier();
#pragma unroll
	for (int i = 0; i < AV; ++i) {
		const int a = tr + TR * i;
		cplx u[WV];
#pragma unroll
		for (int kb = 0; kb < WV; ++kb) RowMap{ rbase + kb * PITCH }.load(data, a, u[kb]);
		dftR<WV, true>(u);
#pragma unroll
		for (int b = 0; b < WV; ++b) W[a + 1024 * b] = u[b];
	}
}

// ------------------------------------------------------------------ small calls: partitioned head with a delay line
//
// One row of NF = 2 B points per pair: window = [previous block | this block] of the pair's ring, forward transform (kept:
// it is the newest entry of the pair's delay line), Y = sum_p X[now - p] H_p over the P1 head partitions, inverse transform,
// the last B points are this block's outputs; the overlap-save convolver's share of the filter (taps from P1 B on, computed
// once per P1 blocks: conv.cpp) is added from `tail` on the way out.  Every thread owns the same 16 bins in every
// transform, so the delay line is only ever re-read by the thread that wrote it: consecutive sub-blocks of one call run in
// ONE launch without any global synchronisation.  Traffic per pair and block: 16 B x NF in, 16 B x NF x P1 delay line,
// 16 B x B out -- the reference's plan for 65536 taps re-reads 59 partitions' worth per block, this one P1 = 8.
template <int LOG2NF>
__global__ __launch_bounds__(NT, 2) void conv_fdl(FdlParams p)
{
	using Cfg = RowCfg<LOG2NF>;
	constexpr int NF = Cfg::N2, P = Cfg::P, RPW = Cfg::RPW, B = NF / 2;
	constexpr bool WL = Cfg::WAVE_LOCAL;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	cplx *data = reinterpret_cast<cplx *>(smem_raw);
	cplx *t256 = data + RPW * Cfg::PITCH, *tlo = t256 + Cfg::T256, *thi = tlo + Cfg::TLO;
	const int tid = threadIdx.x;
	const int rw = tid / P, j = tid % P;
	// XCD-aware order: the pairs of one stream write neighbouring 16-byte pieces of the same output lines, so they should
	// share an L2: workgroup ids that differ by a multiple of 8 land on the same XCD
	long pair = (long) blockIdx.x * RPW + rw;
	if (RPW == 1 && !p.spec_out && p.pairs_per_stream > 1 && (p.n_pairs % (8L * p.pairs_per_stream)) == 0) {
		const long per = 8L * p.pairs_per_stream, blk = blockIdx.x / per, r = blockIdx.x % per;
		pair = (blk * 8 + r % 8) * p.pairs_per_stream + r / 8;
	}
	t256[twpad(tid)] = p.tw_nf[tid * (NF / 256)];
	if (tid < 64) tlo[twpad(tid)] = p.tw_nf[tid];
	else if (tid < 64 + NF / 64) thi[tid - 64] = p.tw_nf[(tid - 64) * 64];
	const bool active = pair < p.n_pairs;
	const cplx *ring = p.ring + (active ? pair : 0) * p.ring_row_stride;
	const TwRow<NF> tw{ t256, tlo, thi };
	const RowMap map{ rw * Cfg::PITCH };
	const long s = (active ? pair : 0) / p.pairs_per_stream;
	const int qs = (int) ((active ? pair : 0) % p.pairs_per_stream);
	const int cha = p.pair_out_ch ? p.pair_out_ch[2 * qs] : -1, chb = p.pair_out_ch ? p.pair_out_ch[2 * qs + 1] : -1;
	double *out = p.out + (size_t) s * p.out_stride_frames * p.C;
	const double *tail = p.tail ? p.tail + ((size_t) s * p.tail_stride_frames + p.tail_off) * p.C : nullptr;
	const bool wide = (chb == cha + 1) && ((cha & 1) == 0) && ((p.C & 1) == 0) && ((((size_t) out) & 15) == 0) && (!tail || (((size_t) tail) & 15) == 0);
	const size_t slot_stride = (size_t) p.n_pairs * NF;
	// Rows of at least 1024 points belong to whole waves (P >= 64): the pair is wave-uniform and every array is addressed as
	// SGPR base (buffer descriptor) + ONE per-lane offset register + a scalar offset per access.  With plain pointers the 16
	// strided accesses per array (4 KB and more apart: beyond the instruction's immediate) each hold a 64-bit address pair:
	// 87 scratch instructions at NF = 4096, two workgroups per CU.
	constexpr bool BUF = (P >= 64);
	typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
	const long pair_u = BUF ? (((long) __builtin_amdgcn_readfirstlane((int) ((active ? pair : 0) >> 32)) << 32) | (unsigned) __builtin_amdgcn_readfirstlane((int) (active ? pair : 0))) : 0;
	constexpr int RSRC_FLAGS = 0x00020000;                      // raw buffer, 32-bit offsets
	const __amdgpu_buffer_rsrc_t r_fdl = __builtin_amdgcn_make_buffer_rsrc(p.fdl + (BUF ? pair_u * NF : 0), 0, 0x7fffffff, RSRC_FLAGS);
	const __amdgpu_buffer_rsrc_t r_H = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx *>(p.Hf), 0, 0x7fffffff, RSRC_FLAGS);
	const __amdgpu_buffer_rsrc_t r_ring = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx *>(p.ring) + (BUF ? pair_u * p.ring_row_stride : 0), 0, 0x7fffffff, RSRC_FLAGS);
	// (32-bit byte offsets: the host only enters this regime when the delay line of a pair's slots and a ring row stay below 2 GB)
	const int jb = j * 16;
	const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out + (BUF ? (size_t) (pair_u / p.pairs_per_stream) * p.out_stride_frames * p.C : 0), 0, 0x7fffffff, RSRC_FLAGS);
	const __amdgpu_buffer_rsrc_t r_tail = __builtin_amdgcn_make_buffer_rsrc(
		const_cast<double *>(p.tail ? p.tail + ((BUF ? (size_t) (pair_u / p.pairs_per_stream) : 0) * p.tail_stride_frames + p.tail_off) * p.C : p.out), 0, 0x7fffffff, RSRC_FLAGS);
	const bool out_small = (double) p.out_stride_frames * p.C * 8 < 2.0e9 && (double) p.n_sub * B * p.C * 8 < 2.0e9;
	const int vo_out = (j * p.C + cha) * 8;
	__syncthreads();
	for (int b = 0; b < p.n_sub; ++b) {
		cplx v[16];
		const long w0 = p.win_base + (long) b * B;
		if constexpr (BUF) {
#pragma unroll
			for (int m = 0; m < 16; ++m)
				v[m] = active ? __builtin_bit_cast(cplx, __builtin_amdgcn_raw_buffer_load_b128(r_ring, (int) ((w0 + j + P * m) & p.ring_mask) * 16, 0, 0)) : make_double2(0.0, 0.0);
		}
		else {
#pragma unroll
			for (int m = 0; m < 16; ++m) v[m] = active ? ring[(w0 + j + P * m) & p.ring_mask] : make_double2(0.0, 0.0);
		}
		if (b > 0) row_sync<WL>();       // the previous sub-block's last gather is done
		row_fft<LOG2NF, false>(v, j, data, map, tw);
		if (p.spec_out) {
			if (active) {
#pragma unroll
				for (int m = 0; m < 16; ++m) p.spec_out[(size_t) pair * NF + j + P * m] = make_double2(v[m].x * p.h_scale, v[m].y * p.h_scale);
			}
			return;
		}
		const int slot = (p.slot0 + b) % p.P1;
		if constexpr (BUF) {
			const int so = (int) ((size_t) slot * slot_stride * 16);
			if (active) {
#pragma unroll
				for (int m = 0; m < 16; ++m) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[m]), r_fdl, jb + so + P * m * 16, 0, 0);
			}
#pragma unroll
			for (int m = 0; m < 16; ++m) v[m] = cmul(v[m], __builtin_bit_cast(cplx, __builtin_amdgcn_raw_buffer_load_b128(r_H, jb, P * m * 16, 0)));
			if (active) {
				for (int q = 1; q < p.P1; ++q) {
					const int sl = (slot + p.P1 - q) % p.P1;
					const int sx = (int) ((size_t) sl * slot_stride * 16), sh = q * NF * 16;
#pragma unroll
					for (int half = 0; half < 2; ++half) {
						cplx x[8];
#pragma unroll
						for (int m = 0; m < 8; ++m) x[m] = __builtin_bit_cast(cplx, __builtin_amdgcn_raw_buffer_load_b128(r_fdl, jb, sx + P * (8 * half + m) * 16, 0));
#pragma unroll
						for (int m = 0; m < 8; ++m) {
							const cplx h = __builtin_bit_cast(cplx, __builtin_amdgcn_raw_buffer_load_b128(r_H, jb, sh + P * (8 * half + m) * 16, 0));
							cplx &a = v[8 * half + m];
							a.x = fma(x[m].x, h.x, fma(-x[m].y, h.y, a.x));
							a.y = fma(x[m].x, h.y, fma(x[m].y, h.x, a.y));
						}
					}
				}
			}
		}
		else {
			cplx *line = p.fdl + (size_t) pair * NF + j;
			if (active) {
#pragma unroll
				for (int m = 0; m < 16; ++m) line[(size_t) slot * slot_stride + P * m] = v[m];
			}
			{
				const cplx *H = p.Hf + j;
#pragma unroll
				for (int m = 0; m < 16; ++m) v[m] = cmul(v[m], H[P * m]);
			}
			if (active) {
				for (int q = 1; q < p.P1; ++q) {
					const int sl = (slot + p.P1 - q) % p.P1;
					const cplx *X = line + (size_t) sl * slot_stride;
					const cplx *H = p.Hf + (size_t) q * NF + j;
					// (eight bins at a time: two workgroups per CU need the kernel inside 256 registers)
#pragma unroll
					for (int half = 0; half < 2; ++half) {
						cplx x[8];
#pragma unroll
						for (int m = 0; m < 8; ++m) x[m] = X[P * (8 * half + m)];
#pragma unroll
						for (int m = 0; m < 8; ++m) {
							const cplx h = H[P * (8 * half + m)];
							cplx &a = v[8 * half + m];
							a.x = fma(x[m].x, h.x, fma(-x[m].y, h.y, a.x));
							a.y = fma(x[m].x, h.y, fma(x[m].y, h.x, a.y));
						}
					}
				}
			}
		}
		row_sync<WL>();
		row_fft<LOG2NF, true>(v, j, data, map, tw);
		if (!active) continue;
		// outputs: positions B .. NF - 1 of the row = frames b B .. b B + B - 1 of this launch
#pragma unroll
		for (int m = 8; m < 16; ++m) {
			const long f = (long) b * B + (j + P * m - B);
			cplx y = v[m];
			if constexpr (BUF) {
				if (wide && out_small) {
					// frame f = (b - 1) B + j + P m: one per-lane offset (j, the pair's channels), the rest scalar
					const int so = (int) ((((long) b - 1) * B + P * m) * p.C * 8);
					if (tail) { const cplx t = __builtin_bit_cast(cplx, __builtin_amdgcn_raw_buffer_load_b128(r_tail, vo_out, so, 0)); y.x += t.x; y.y += t.y; }
					__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), r_out, vo_out + so, 0, 0);
					continue;
				}
			}
			if (wide) {
				if (tail) { const cplx t = *reinterpret_cast<const cplx *>(tail + f * p.C + cha); y.x += t.x; y.y += t.y; }
				*reinterpret_cast<cplx *>(out + f * p.C + cha) = y;
			}
			else {
				if (cha >= 0) out[f * p.C + cha] = y.x + (tail ? tail[f * p.C + cha] : 0.0);
				if (chb >= 0) out[f * p.C + chb] = y.y + (tail ? tail[f * p.C + chb] : 0.0);
			}
		}
	}
}

template <int L> static void launch_fdl_t(const FdlParams &p, hipStream_t st)
{
	using Cfg = RowCfg<L>;
	grant_lds(conv_fdl<L>, Cfg::LDS);
	const long wgs = (p.n_pairs + Cfg::RPW - 1) / Cfg::RPW;
	hipLaunchKernelGGL(conv_fdl<L>, dim3((unsigned) wgs), dim3(NT), Cfg::LDS, st, p);
}

void launch_conv_fdl(const FdlParams &p, hipStream_t st)
{
	switch (p.log2NF) {
	case 9: launch_fdl_t<9>(p, st); break;
	case 10: launch_fdl_t<10>(p, st); break;
	case 11: launch_fdl_t<11>(p, st); break;
	case 12: launch_fdl_t<12>(p, st); break;
	default: break;
	}
}

// interleaved slab -> pair rings for the selected channels (+ pass-through of the others to `out`).
// Adjacent lanes hold adjacent channels of a frame, so the two 8-byte halves of a ring element leave together.
__global__ __launch_bounds__(NT) void conv_deinterleave(DeintParams p)
{
	const int s = blockIdx.y;
	const long in0 = (long) s * p.in_stride_frames * p.C;
	const double *in = p.in + in0;
	double *out = p.out ? p.out + (size_t) s * p.out_stride_frames * p.C : nullptr;
	double *ring = reinterpret_cast<double *>(p.ring);
	const long n = p.frames * p.C;
	const bool wire_in = p.in_fmt != PCM_DOUBLE;     // the first kernel of a pipeline fed in a wire format (any of them)
	for (long e = (long) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long) gridDim.x * blockDim.x) {
		const long t = e / p.C;
		const int c = (int) (e - t * p.C);
		const int slot = p.slot_of_channel[c];
		double v = wire_in ? pcm_load(p.in, p.in_fmt, in0 + e) : in[e];
		if (slot >= 0) {
			if (p.round_f32) v = (double) (float) v;
			ring[2 * (((size_t) s * p.rows_per_stream + (slot >> 1)) * p.ring_row_stride + ((p.pos + t) & p.ring_mask)) + (slot & 1)] = v;
		}
		else if (out) out[e] = v;
	}
}

// direct-form FIR for <= 32 taps (fir.c:43-62, fir_p.c:131-148), bit-exact: the reference scatter-adds each input
// into a circular accumulator in time order, i.e. every output is ((0 + x[j-T+1] h[T-1]) + ... ) + x[j] h[0]
// with separately rounded products and sums.
__global__ __launch_bounds__(NT) void fir_direct_kernel(FirDirectParams p)
{
	// bit-exact contract: every product and every sum rounded on its own, as the reference's x86-64 build does.  hipcc
	// contracts a * b + c into one fused operation by default -- also through __dmul_rn / __dadd_rn, which are plain
	// operators to the optimiser (round 1 shipped that: outputs one ulp off the reference in a third of the samples)
#pragma clang fp contract(off)
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
			const double prod = x * h[m];      // (plain operators: the pragma above governs them, not the bodies of inlined helpers)
			acc = acc + prod;
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

template <class K> static void grant_lds(K kernel, size_t bytes) { grant_dynamic_lds(reinterpret_cast<const void *>(kernel), bytes); }

template <int L> static void launch_col_fwd(const ConvParams &p, int n_pairs, hipStream_t st)
{
	using Cfg = ColCfg<L, 1>;
	if (p.slab && p.slab_fmt != PCM_DOUBLE) {
		grant_lds(conv_col_fwd<L, true>, Cfg::LDS);
		hipLaunchKernelGGL((conv_col_fwd<L, true>), dim3((unsigned) (p.N2 / Cfg::TW), n_pairs), dim3(NT), Cfg::LDS, st, p);
		return;
	}
	grant_lds(conv_col_fwd<L, false>, Cfg::LDS);
	hipLaunchKernelGGL((conv_col_fwd<L, false>), dim3((unsigned) (p.N2 / Cfg::TW), n_pairs), dim3(NT), Cfg::LDS, st, p);
}

template <int L, int PPS> static void launch_col_inv_pps(const ConvParams &p, hipStream_t st)
{
	using Cfg = ColCfg<L, PPS>;
	const int groups = (p.pairs_per_stream + PPS - 1) / PPS;
	const dim3 grid((unsigned) (p.N2 / Cfg::TW), (unsigned) (p.n_streams_launch * groups)), block(Cfg::THREADS);
	if (PPS == 4 && p.nph == 2 && p.up == 2 && p.down == 1 && !p.round_f32 && !p.ring_out_round_f32) {
		if constexpr (PPS == 4) {
			grant_lds(conv_col_inv<L, PPS, 2>, Cfg::LDS);
			hipLaunchKernelGGL((conv_col_inv<L, PPS, 2>), grid, block, Cfg::LDS, st, p);
			return;
		}
	}
	if (p.nph == 1 && p.up == 1 && p.down == 1) {
		grant_lds(conv_col_inv<L, PPS, 0>, Cfg::LDS);
		hipLaunchKernelGGL((conv_col_inv<L, PPS, 0>), grid, block, Cfg::LDS, st, p);
		return;
	}
	grant_lds(conv_col_inv<L, PPS, 1>, Cfg::LDS);
	hipLaunchKernelGGL((conv_col_inv<L, PPS, 1>), grid, block, Cfg::LDS, st, p);
}

template <int L> static void launch_col_inv(const ConvParams &p, hipStream_t st)
{
	if (p.pairs_per_stream >= 3) launch_col_inv_pps<L, 4>(p, st);
	else if (p.pairs_per_stream == 2) launch_col_inv_pps<L, 2>(p, st);
	else launch_col_inv_pps<L, 1>(p, st);
}

void launch_conv_col(const ConvParams &p, bool inverse, int n_pairs, hipStream_t st)
{
	switch (p.log2N1) {
	case 4: if (inverse) launch_col_inv<4>(p, st); else launch_col_fwd<4>(p, n_pairs, st); break;
	case 5: if (inverse) launch_col_inv<5>(p, st); else launch_col_fwd<5>(p, n_pairs, st); break;
	case 6: if (inverse) launch_col_inv<6>(p, st); else launch_col_fwd<6>(p, n_pairs, st); break;
	case 7: if (inverse) launch_col_inv<7>(p, st); else launch_col_fwd<7>(p, n_pairs, st); break;
	case 8: if (inverse) launch_col_inv<8>(p, st); else launch_col_fwd<8>(p, n_pairs, st); break;
	default: break;
	}
}

template <int L2> static void launch_row(const ConvParams &p, int mode, int n_pairs, hipStream_t st)
{
	using Cfg = RowCfg<L2>;
	if (mode == 1) grant_lds(conv_row<L2, 1>, Cfg::LDS); else if (mode == 2) grant_lds(conv_row<L2, 2>, Cfg::LDS); else grant_lds(conv_row<L2, 0>, Cfg::LDS);
	dim3 grid((unsigned) (p.N1 / Cfg::RPW), n_pairs), block(NT);
	if (mode == 1) hipLaunchKernelGGL((conv_row<L2, 1>), grid, block, Cfg::LDS, st, p);
	else if (mode == 2) hipLaunchKernelGGL((conv_row<L2, 2>), grid, block, Cfg::LDS, st, p);
	else hipLaunchKernelGGL((conv_row<L2, 0>), grid, block, Cfg::LDS, st, p);
}

template <int WV> static void launch_row_big(const ConvParams &p, int mode, int n_pairs, hipStream_t st)
{
	constexpr size_t LDS = ((size_t) 4 * RowCfg<10>::PITCH + RowCfg<10>::T256 + 3 * 64 + RowCfg<10>::TLO) * sizeof(cplx);
	if (mode) grant_lds(conv_row_big<WV, 1>, LDS); else grant_lds(conv_row_big<WV, 0>, LDS);
	dim3 grid((unsigned) (p.N1 / (4 / WV)), n_pairs), block(NT);
	if (mode == 1) hipLaunchKernelGGL((conv_row_big<WV, 1>), grid, block, LDS, st, p);
	else hipLaunchKernelGGL((conv_row_big<WV, 0>), grid, block, LDS, st, p);
}

static void pipe_grid(int groups, int n_pairs, int *ranges, int *per)
{
	// one workgroup per CU (LDS): the row groups times as many pair ranges as it takes to give every CU two workgroups in turn
	static const int wgs = [] { const char *e = getenv("DSP_AMD_ROW_PIPE_WGS"); return e ? atoi(e) : 512; }();
	int r = (wgs + groups - 1) / groups;
	if (r > n_pairs) r = n_pairs;
	if (r < 1) r = 1;
	*per = (n_pairs + r - 1) / r;
	*ranges = (n_pairs + *per - 1) / *per;
}

template <int L2> static void launch_row_pipe(const ConvParams &p, int n_pairs, hipStream_t st)
{
	using Cfg = RowCfg<L2>;
	constexpr size_t LDS = ((size_t) 16 * NT + (size_t) Cfg::RPW * Cfg::PITCH + Cfg::NTW) * sizeof(cplx);
	const int groups = (int) (p.N1 / Cfg::RPW);
	int ranges, per;
	pipe_grid(groups, n_pairs, &ranges, &per);
	if (p.nph == 2) {
		grant_lds((conv_row_pipe<L2, 2>), LDS);
		hipLaunchKernelGGL((conv_row_pipe<L2, 2>), dim3((unsigned) groups, (unsigned) ranges), dim3(NT), LDS, st, p, per, n_pairs);
		return;
	}
	grant_lds((conv_row_pipe<L2, 1>), LDS);
	hipLaunchKernelGGL((conv_row_pipe<L2, 1>), dim3((unsigned) groups, (unsigned) ranges), dim3(NT), LDS, st, p, per, n_pairs);
}

template <int L2> static void launch_row_duo(const ConvParams &p, int n_pairs, hipStream_t st)
{
	using Cfg = RowCfg<L2>;
	constexpr size_t LDS = ((size_t) Cfg::RPW * Cfg::PITCH + Cfg::NTW) * sizeof(cplx);
	const int groups = (int) (p.N1 / Cfg::RPW);
	// two workgroups per CU: the row groups times as many pair ranges as it takes to fill 512 slots
	int r = (512 + groups - 1) / groups;
	if (r > n_pairs) r = n_pairs;
	if (r < 1) r = 1;
	const int per = (n_pairs + r - 1) / r, ranges = (n_pairs + per - 1) / per;
	grant_lds((conv_row_duo<L2>), LDS);
	hipLaunchKernelGGL((conv_row_duo<L2>), dim3((unsigned) groups, (unsigned) ranges), dim3(NT), LDS, st, p, per, n_pairs);
}

// Which row-kernel family serves a plan -- decided from the plan alone (never from the number of pairs in a launch): the
// filter spectra are stored in the family's own order by its preparation mode.
//   persistent three-pass kernel (conv_row_pipe; conv_row for launches of a few pairs): single-phase plans with one shared filter;
//   split rows (conv_row_big: the row FFT itself four-step, all but one exchange inside a wave): the other 2048-point plans
//   (a persistent form of it was measured: 5.4 against 4.4 ms at 2048-point rows, 11.1 against 8.9 at 4096 -- its two extra
//   exchanges per row are exposed when a SIMD holds a single wave);
//   everything else the generic three-pass kernel
static const int g_pipe_env = [] { const char *e = getenv("DSP_AMD_ROW_PIPE"); return e ? atoi(e) : 1; }();
static bool plan_is_pipe(const ConvParams &p) { return g_pipe_env && p.nph <= 2 && p.shared_h; }
static bool rows_are_split(const ConvParams &p)
{
	static const int big_env = [] { const char *e = getenv("DSP_AMD_ROW_BIG"); return e ? atoi(e) : 1; }();
	if (!big_env || p.nph > 1 || plan_is_pipe(p)) return false;          // (a multi-phase plan uses the generic kernel for preparation too)
	return p.log2N2 == 11 || (big_env > 1 && p.log2N2 == 12);            // (measured: the 3-pass kernel is ahead at 4096)
}

void launch_conv_row(const ConvParams &p, int mode, int n_pairs, hipStream_t st)
{
	if (rows_are_split(p)) {
		if (p.log2N2 == 11) launch_row_big<2>(p, mode, n_pairs, st); else launch_row_big<4>(p, mode, n_pairs, st);
		return;
	}
	// (the two-branch form holds both filter rows in registers: at 2048- / 4096-point rows it spills 100 VGPRs and is behind the
	// one-shot kernel, 16.7 against 15.0 ms; at 1024-point rows ahead, 3.26 against 3.53)
	static const int duo_env = [] { const char *e = getenv("DSP_AMD_ROW_DUO"); return e ? atoi(e) : 1; }();
	if (duo_env && plan_is_pipe(p) && mode == 0 && p.nph == 1 && n_pairs >= 8 && p.log2N2 >= 11 && (duo_env == 1 || p.log2N2 == 10 + duo_env)) {
		if (p.log2N2 == 11) launch_row_duo<11>(p, n_pairs, st); else launch_row_duo<12>(p, n_pairs, st);
		return;
	}
	if (plan_is_pipe(p) && (mode == 0 || (mode == 2 && p.nph == 2)) && n_pairs >= 8) {
		switch (p.log2N2) {
		case 9: launch_row_pipe<9>(p, n_pairs, st); return;
		case 10: launch_row_pipe<10>(p, n_pairs, st); return;
		case 11: launch_row_pipe<11>(p, n_pairs, st); return;
		case 12: launch_row_pipe<12>(p, n_pairs, st); return;
		default: break;
		}
	}
	switch (p.log2N2) {
	case 9: launch_row<9>(p, mode, n_pairs, st); break;
	case 10: launch_row<10>(p, mode, n_pairs, st); break;
	case 11: launch_row<11>(p, mode, n_pairs, st); break;
	case 12: launch_row<12>(p, mode, n_pairs, st); break;
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
