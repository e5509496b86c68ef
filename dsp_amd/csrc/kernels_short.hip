// kernels_short.hip -- one-trip FFT convolution for SHORT filters behind long calls (round 5).
//
// What it replaces, per block, in the reference: fir_p's / fir's block transform, spectrum product and inverse (fir_p.c:64-103, fir.c:109-149) for
// filters of up to 4097 taps -- `hilbert -p 4095` (hilbert.c:28-92), crossover and correction FIRs, the FIRs `biquad -r` sections are designed into.
// The four-step convolver (kernels_fft.hip) sends every window through HBM three times whatever the filter's length; a window of 8192 points of a
// channel pair is 128 KB and fits one workgroup's LDS, so here a block is ONE read of the window (first_n frames of history + hop new ones) and ONE
// write of hop outputs: (N + hop) / hop units of 16 bytes per pair and frame -- 3 at 4095 taps, 2.3 at 1000 -- against 6.4 for three trips of a
// 65536-point transform.  BASELINE config 5's `hilbert -p 4095` stage was 22 of its 35 ms.
//
// What bounds it (BASELINE config 5's hilbert stage: 1024 pairs x 224 blocks, 44 GB): not memory -- a block is two 8192-point transforms for 4096
// outputs, about 7.5 us of fp64 issue and 6 us of LDS exchange traffic per CU against 9 us of HBM time.  Counters (profiles/r06_conv_short_config5_
// counters.json, round 5's one-workgroup form): waves parked at a barrier or a counter 42 % of their time, issuing VALU 27 %, stalled on an
// instruction's operands 20 %, LDS bank conflicts 18 % of the LDS cycles.  Round 5: 15.6 - 17.5 ms; with the per-lane descriptors gone (16 loops per
// block over one value) 16.4 - 16.7; round 6's form below 15.9.  A prefetched next window measured slower (profiles/r05_conv_short_prefetch_ab.txt).
//
// Workgroup = one channel pair (z = x_a + i x_b, exact: h is real), 256 threads x two sets of 16 points, walking the pair's blocks; radix 16 / 16 / 16 / 2
// Stockham passes whose exchanges go through a buffer of 8192 DOUBLES -- real parts, then imaginary parts -- so that two independent workgroups fit a CU
// (72 KB of LDS each) and one issues butterflies while the other waits; conflict-free slots for both access shapes (short_fft2).  The filter row comes from
// L2 where it is used.  Same ring / slab / output conventions as K1 and K3 (fft_params.h: ShortParams).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
#include "kparams.h"
#include "fft_params.h"
#include "pcm_device.h"

namespace dspamd {
namespace psh {
typedef double real;
#define FFT_F32 0
#define FFT_CORE_NO_LAUNCHERS 1
#include "fft_core.inc"
#undef FFT_CORE_NO_LAUNCHERS
#undef FFT_F32

// Geometry of an instance: N = 2^LOG2N points of a channel pair, N / 32 threads that hold TWO sets of 16 points each (the points of "virtual threads" j and
// j + N / 32: v[t][m] <-> position j + (N / 32) (t + 2 m), the 32 inputs of the thread's radix-32 butterfly in EVERY pass), a direct twiddle table for the
// middle pass (W_512 / W_1024) and the lo / hi pair for W_N.
template <int LOG2N> struct ShCfg {
	static constexpr int N = 1 << LOG2N, P = N / 16, NTH = N / 32;
	static constexpr int TD = N / 16, TDP = TD + TD / 16, TLO = 68, THI = N / 64;
	static constexpr size_t LDS = (size_t) N * sizeof(double) + ((size_t) TDP + TLO + THI) * sizeof(cplx);
};
// get<M>(e) = exp(-2 pi i e / M): M = N as hi[e >> 6] * lo[e & 63], M <= TD from the direct table (twpad: power-of-two strides conflict-free)
template <int NSEQ, int TD> struct TwShort {
	const cplx *td, *lo, *hi;
	template <int M> __device__ __forceinline__ cplx get(int e) const
	{
		if constexpr (M == NSEQ) return cmul(hi[e >> 6], lo[twpad(e & 63)]);
		else { static_assert(M <= TD, "no table for this pass"); return td[twpad(e * (TD / M))]; }
	}
};

// The exchanges carry the real and the imaginary parts one after the other through a buffer of N doubles: twice the barriers and LDS instructions for the
// same bytes -- but half the LDS, so that two independent 8192-point workgroups fit a CU (one issues butterflies while the other waits), and a 16384-point
// window fits a CU at all.  Slots: an element is 8 bytes = 2 banks of 64; a ds_read / ds_write_b64 is served in two groups of 32 lanes, conflict-free when
// the 32 slots differ mod 32: slot = pos ^ ((pos >> 5) & 31) makes them for every access shape of the passes below (stores 32 j + r, (j - k) 32 + k + 32 r,
// (j - k) 16 + k + 32 r; gathers j + (N / 32) m -- simulated over every pass before it ran).
__device__ __forceinline__ int sh_slotd(int pos) { return pos ^ ((pos >> 5) & 31); }

// One radix-32 Stockham pass on the thread's 32 points, u[2 m] = va[m], u[2 m + 1] = vb[m]: butterfly b = j, stride NS.  32 = 2 x 16: the two sets'
// 16-point transforms, w_32^k2 on the odd set's results (constants), a radix-2 step across the sets.  Output r of the butterfly ends up in va[r] (r < 16) or
// vb[r - 16]; its Stockham position is (j - k) 32 + k + NS r, k = j mod NS (sh_pos32).
template <int NS, bool INV, class Tw>
__device__ __forceinline__ void sh_pass32(cplx (&va)[16], cplx (&vb)[16], int j, const Tw &tw)
{
	if constexpr (NS > 1) {
		const int k = j & (NS - 1);
#pragma unroll
		for (int m = 0; m < 16; ++m) {
			if (m) { const cplx w = tw.template get<32 * NS>(2 * m * k); va[m] = INV ? cmulc(va[m], w) : cmul(va[m], w); }
			const cplx w = tw.template get<32 * NS>((2 * m + 1) * k);
			vb[m] = INV ? cmulc(vb[m], w) : cmul(vb[m], w);
		}
	}
	dft16<INV>(va);
	dft16<INV>(vb);
	constexpr real c1 = 0.98078528040323044913, s1 = 0.19509032201612826785, c2 = 0.92387953251128675613, s2 = 0.38268343236508977173;
	constexpr real c3 = 0.83146961230254523708, s3 = 0.55557023301960222474, h = 0.70710678118654752440;
	// w_32^k2 = (cos(pi k2 / 16), -sin(pi k2 / 16))
	vb[1] = mul_w<INV>(vb[1], c1, s1);   vb[2] = mul_w<INV>(vb[2], c2, s2);    vb[3] = mul_w<INV>(vb[3], c3, s3);
	vb[4] = INV ? mkc((vb[4].x - vb[4].y) * h, (vb[4].x + vb[4].y) * h) : mkc((vb[4].x + vb[4].y) * h, (vb[4].y - vb[4].x) * h);
	vb[5] = mul_w<INV>(vb[5], s3, c3);   vb[6] = mul_w<INV>(vb[6], s2, c2);    vb[7] = mul_w<INV>(vb[7], s1, c1);
	vb[8] = mul_mi<INV>(vb[8]);
	vb[9] = mul_w<INV>(vb[9], -s1, c1);  vb[10] = mul_w<INV>(vb[10], -s2, c2); vb[11] = mul_w<INV>(vb[11], -s3, c3);
	vb[12] = INV ? mkc(-(vb[12].x + vb[12].y) * h, (vb[12].x - vb[12].y) * h) : mkc((vb[12].y - vb[12].x) * h, -(vb[12].x + vb[12].y) * h);
	vb[13] = mul_w<INV>(vb[13], -c3, s3); vb[14] = mul_w<INV>(vb[14], -c2, s2); vb[15] = mul_w<INV>(vb[15], -c1, s1);
#pragma unroll
	for (int r = 0; r < 16; ++r) { const cplx a = va[r], b = vb[r]; va[r] = cadd(a, b); vb[r] = csub(a, b); }
}
template <int NS> __device__ __forceinline__ int sh_pos32(int j, int r) { const int k = j & (NS - 1); return (j - k) * 32 + k + NS * r; }
template <int NS> __device__ __forceinline__ int sh_pos16(int jv, int r) { const int k = jv & (NS - 1); return (jv - k) * 16 + k + NS * r; }

// The N-point transform of a pair's window: radix 32 / 16 / 16 at 8192 points, 32 / 32 / 16 at 16384 -- TWO exchanges (until round 6's second half: radix
// 16 / 16 / 16 / 2 with three; a thread's 32 points are the same positions in every pass either way, so a radix-32 step costs no exchange of its own).
// Results in natural order at the positions the thread loaded from.
template <int LOG2N, bool INV, class Tw>
__device__ __forceinline__ void short_fft2(cplx (&va)[16], cplx (&vb)[16], int j, double *lds, const Tw &tw)
{
	constexpr int P = ShCfg<LOG2N>::P, H = ShCfg<LOG2N>::NTH;
	const RowMap nomap{ 0 };
	// pa(r), pb(r): Stockham positions of va[r], vb[r]
	auto exchange = [&](auto pa, auto pb, bool last) {
#pragma unroll
		for (int r = 0; r < 16; ++r) { lds[sh_slotd(pa(r))] = va[r].x; lds[sh_slotd(pb(r))] = vb[r].x; }
		lds_barrier();
		double xa[16], xb[16];
#pragma unroll
		for (int m = 0; m < 16; ++m) { xa[m] = lds[sh_slotd(j + P * m)]; xb[m] = lds[sh_slotd(j + H + P * m)]; }
		lds_barrier();
#pragma unroll
		for (int r = 0; r < 16; ++r) { lds[sh_slotd(pa(r))] = va[r].y; lds[sh_slotd(pb(r))] = vb[r].y; }
		lds_barrier();
#pragma unroll
		for (int m = 0; m < 16; ++m) { va[m] = mkc(xa[m], lds[sh_slotd(j + P * m)]); vb[m] = mkc(xb[m], lds[sh_slotd(j + H + P * m)]); }
		if (!last) lds_barrier();            // (behind the last exchange the caller's own barrier stands in front of the next store)
	};
	asm volatile("" : "+v"(j));
	sh_pass32<1, INV>(va, vb, j, tw);
	exchange([&](int r) { return sh_pos32<1>(j, r); }, [&](int r) { return sh_pos32<1>(j, r + 16); }, false);
	asm volatile("" : "+v"(j));
	if constexpr (LOG2N == 13) {
		pass16<LOG2N, 16, 32, INV, true>(va, j, (cplx *) nullptr, nomap, tw);
		pass16<LOG2N, 16, 32, INV, true>(vb, j + H, (cplx *) nullptr, nomap, tw);
		exchange([&](int r) { return sh_pos16<32>(j, r); }, [&](int r) { return sh_pos16<32>(j + H, r); }, true);
	}
	else {
		sh_pass32<32, INV>(va, vb, j, tw);
		exchange([&](int r) { return sh_pos32<32>(j, r); }, [&](int r) { return sh_pos32<32>(j, r + 16); }, true);
	}
	asm volatile("" : "+v"(j));
	pass16<LOG2N, 16, P, INV, true>(va, j, (cplx *) nullptr, nomap, tw);
	pass16<LOG2N, 16, P, INV, true>(vb, j + H, (cplx *) nullptr, nomap, tw);
}

// (every address is a buffer descriptor + a 32-bit offset: the host checks that rings, slabs and outputs stay below 2 GB per stream / pair -- sixteen
// 64-bit address pairs per direction do not fit beside the 64 registers of the window and the 64 of the filter row)
typedef unsigned int sh_u32x2 __attribute__((ext_vector_type(2)));
// BS: bytes per sample of the slab in direct mode (8: fp64; 4: s24 / s32 / float; 2: s16 -- read_buf_<fmt> of pcm_device.h in the loads)
template <int LOG2N, int BS>
__global__ __launch_bounds__(ShCfg<LOG2N>::NTH) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_short(ShortParams p)
{
	typedef ShCfg<LOG2N> Cfg;
	constexpr int N = Cfg::N, P = Cfg::P, NTH = Cfg::NTH, VT = 2;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	double *datad = reinterpret_cast<double *>(smem_raw);                // the exchange buffer: one half (real / imaginary parts) of the row at a time
	cplx *td = reinterpret_cast<cplx *>(smem_raw + (size_t) N * sizeof(double)), *tlo = td + Cfg::TDP, *thi = tlo + Cfg::TLO;
	int j = threadIdx.x;
	const long pair = blockIdx.x;
	const long n_blocks = (p.n_in + p.hop - 1) / p.hop;
	const long b0 = (long) blockIdx.y * p.blocks_per_wg, b1 = (b0 + p.blocks_per_wg < n_blocks) ? b0 + p.blocks_per_wg : n_blocks;
	if (b0 >= b1) return;
	for (int e = j; e < Cfg::TD + 64 + Cfg::THI; e += NTH) {
		if (e < Cfg::TD) td[twpad(e)] = TAB(p.tw)[e * (N / Cfg::TD)];
		else if (e < Cfg::TD + 64) tlo[twpad(e - Cfg::TD)] = TAB(p.tw)[e - Cfg::TD];
		else thi[e - Cfg::TD - 64] = TAB(p.tw)[(e - Cfg::TD - 64) * 64];
	}
	const cplx *Hrow = p.Hout ? nullptr : TAB(p.H) + (long) p.pair_h[pair] * N;
	lds_barrier();                                                       // tables visible
	const TwShort<N, Cfg::TD> tw{ td, tlo, thi };
	// (the division runs on the vector unit; its result is uniform all the same and is said to be: with a per-lane stream index the slab and output
	// descriptors are per-lane values and every load through them becomes a loop over their distinct values -- 16 such loops per block until round 6)
	const long s = __builtin_amdgcn_readfirstlane((int) (pair / p.pairs_per_stream)), qs = pair - s * p.pairs_per_stream;
	const int fb = p.C * (int) sizeof(double);                          // bytes per fp64 output frame
	const int fbi = p.C * BS;                                           // bytes per slab frame
	const WordFormat wf_slab = word_format(p.slab_fmt);
	const int mask = (int) p.ring_mask, omask = (int) p.ring_out_mask;
	const __amdgpu_buffer_rsrc_t r_ring = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2 *>(p.ring) + pair * p.ring_row_stride, 0, rsrc_records((p.ring_mask + 1) * 16), 0x00020000);
	// direct mode: the pair's two channels of a slab frame are 16 contiguous bytes (channels 2 qs, 2 qs + 1: the host checked)
	const __amdgpu_buffer_rsrc_t r_slab = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(p.slab ? reinterpret_cast<const char *>(p.slab) + ((size_t) s * p.slab_stride_frames * p.C + 2 * qs) * BS : nullptr), 0,
		p.slab ? rsrc_records((p.slab_frames * p.C - 2 * qs) * BS) : 0, 0x00020000);
	auto slab_ld = [&](int vo) -> cplx {
		if constexpr (BS == 8) return buf_ldc(r_slab, vo, 0);
		else if constexpr (BS == 4) { const sh_u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(r_slab, vo, 0, 0); return mkc(pcm_from_word(w.x, wf_slab), pcm_from_word(w.y, wf_slab)); }
		else { const uint32_t w = __builtin_amdgcn_raw_buffer_load_b32(r_slab, vo, 0, 0); return mkc(pcm_from_s16(w & 0xffffu), pcm_from_s16(w >> 16)); }
	};
	const int cha = p.pair_out_ch[2 * qs], chb = p.pair_out_ch[2 * qs + 1];
	double *out = p.out ? p.out + (size_t) s * p.out_stride_frames * p.C : nullptr;
	const bool wide = (chb == cha + 1) && ((cha & 1) == 0) && ((p.C & 1) == 0) && ((((size_t) out) & 15) == 0);
	const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, (out && !p.sink.on) ? rsrc_records(p.out_count * p.C * (long) sizeof(double)) : 0, 0x00020000);
	const __amdgpu_buffer_rsrc_t r_rout = __builtin_amdgcn_make_buffer_rsrc(p.ring_out ? p.ring_out + pair * p.ring_out_stride : nullptr, 0, p.ring_out ? rsrc_records((p.ring_out_mask + 1) * 16) : 0, 0x00020000);
	const int first_n = (int) p.first_n;
	for (long b = b0; b < b1; ++b) {
		const long q_blk = p.q0 + b * p.hop;
		const int in_count = (int) ((p.n_in - b * p.hop < p.hop) ? p.n_in - b * p.hop : p.hop);
		const long a0 = q_blk - p.lat - p.first_n;                  // input index of window element 0
		const int valid = first_n + in_count;
		const int r0 = (int) (a0 & p.ring_mask);                    // its ring position
		// window coordinates at which the slab starts / from which slab frames are filed in the ring (N: never)
		const long d_slab = p.slab_q0 - a0, d_file = p.file_from - a0;
		const int n_slab = !p.slab ? N : (d_slab <= 0 ? 0 : (d_slab < N ? (int) d_slab : N));
		const int n_file = (d_file <= first_n) ? first_n : (d_file < N ? (int) d_file : N);
		const int so = (int) (-d_slab * fbi);                        // byte offset of element 0 in the stream's slab (negative while it lies in older calls)
		cplx v[VT][16];
		asm volatile("" : "+v"(j));                                  // (addresses are recomputed per block: kept across blocks they are 40 registers nobody has)
#pragma unroll
		for (int t = 0; t < VT; ++t) {
			const int jv = j + NTH * t;
#pragma unroll
			for (int m = 0; m < 16; ++m) {
				const int n = jv + P * m;
				if (n >= valid) v[t][m] = mkc(0.0, 0.0);
				else if (n >= n_slab) {
					v[t][m] = slab_ld(so + n * fbi);
					if (n >= n_file) buf_stc(v[t][m], r_ring, ((r0 + n) & mask) * 16);
				}
				else v[t][m] = buf_ldc(r_ring, ((r0 + n) & mask) * 16, 0);
			}
		}
		if (b > b0) lds_barrier();                                   // the previous block's last gather is done
		short_fft2<LOG2N, false>(v[0], v[1], j, datad, tw);
		if (p.Hout) {
#pragma unroll
			for (int t = 0; t < VT; ++t) {
				cplx *Ho = reinterpret_cast<cplx *>(p.Hout) + pair * N + j + NTH * t;
#pragma unroll
				for (int m = 0; m < 16; ++m) Ho[P * m] = mkc(v[t][m].x * p.h_scale, v[t][m].y * p.h_scale);
			}
			return;
		}
		{
			// (no room for the filter row beside two sets of points: it comes from L2 where it is used -- the same 128 KB for every pair of a filter;
			// asked for HERE, from an index the compiler cannot see through: hoisted above the forward transform the loads are 32 spilled registers)
			int jh = j;
			asm volatile("" : "+v"(jh));
#pragma unroll
			for (int t = 0; t < VT; ++t)
#pragma unroll
				for (int g = 0; g < 2; ++g) {
					cplx hh[8];
#pragma unroll
					for (int m = 0; m < 8; ++m) hh[m] = Hrow[jh + NTH * t + P * (8 * g + m)];
#pragma unroll
					for (int m = 0; m < 8; ++m) v[t][8 * g + m] = cmul(v[t][8 * g + m], hh[m]);
					__builtin_amdgcn_sched_barrier(0);
				}
		}
		lds_barrier();                                               // every gather of the forward transform is done
		short_fft2<LOG2N, true>(v[0], v[1], j, datad, tw);
		// window sample first_n + f -> output frame mo0 + f, for f in [f_lo, f_hi)
		const long mo0 = q_blk - p.k_origin;
		const int f_lo = (mo0 >= 0) ? 0 : (-mo0 < in_count ? (int) -mo0 : in_count);
		const int f_hi = (p.out_count - mo0 >= in_count) ? in_count : (p.out_count - mo0 > 0 ? (int) (p.out_count - mo0) : 0);
		const int ob = (int) ((mo0 * p.C + (cha >= 0 ? cha : 0)) * (long) sizeof(double)), ob2 = (int) ((mo0 * p.C + (chb >= 0 ? chb : 0)) * (long) sizeof(double));
		const int rp0 = (int) ((p.ring_out_pos + mo0) & p.ring_out_mask);
		asm volatile("" : "+v"(j));
		if (p.sink.on) {
			// the END of a pipeline run from wire format to wire format: dither, clip() and the conversion in these stores (dsp.c:685-699), as K3 has
			// them.  A thread's valid outputs are a contiguous range of m, P frames apart: from one to the next the position in the two dither
			// sequences moves by P C samples.
			const bool dither = p.sink.dither_mult != 0.0;
			const int bs = (p.sink.fmt == PCM_DOUBLE) ? 8 : (p.sink.fmt == PCM_S16) ? 2 : 4;
			const WordFormat wf_sink = word_format(p.sink.fmt);
			char *wout = reinterpret_cast<char *>(p.out) + (size_t) s * p.out_stride_frames * p.C * bs;
			const bool wpair = (chb == cha + 1) && ((cha & 1) == 0) && ((p.C & 1) == 0) && ((((size_t) wout) & 15) == 0);
			double peak = 0.0;
			unsigned long long clipped = 0;
			// (a lambda called once per set: as a loop over the sets the body is too large for the unroller, and a rolled loop indexes v by a variable)
			auto sink_set = [&](cplx (&vv)[16], const int jv) {
				int m_first = 16;
#pragma unroll
				for (int m = 15; m >= 0; --m) { const int f = jv + P * m - first_n; if (f >= f_lo && f < f_hi) m_first = m; }
				uint32_t ua0 = 0, ua1 = 0, ub0 = 0, ub1 = 0, j0 = 1, j1 = 1;
				if (dither && m_first < 16) {
					const long mo = mo0 + (jv + P * m_first - first_n);
					const uint64_t na = (uint64_t) (p.sink.samples_before + mo * p.C + (cha >= 0 ? cha : 0)) + 1;
					const uint64_t nb = (uint64_t) (p.sink.samples_before + mo * p.C + (chb >= 0 ? chb : 0)) + 1;
					ua0 = pm_pow<0>(na); ua1 = pm_pow<1>(na);
					if (chb == cha + 1) { ub0 = pm_mul(ua0, PM_A0); ub1 = pm_mul(ua1, PM_A1); }
					else { ub0 = pm_pow<0>(nb); ub1 = pm_pow<1>(nb); }
					j0 = pm_pow<0>((uint64_t) P * p.C); j1 = pm_pow<1>((uint64_t) P * p.C);
				}
#pragma unroll
				for (int m = 0; m < 16; ++m) {
					const int f = jv + P * m - first_n;
					if (f < f_lo || f >= f_hi) continue;
					const long mo = mo0 + f;
					double ya = vv[m].x, yb = vv[m].y;
					if (p.round_f32) { ya = (double) (float) ya; yb = (double) (float) yb; }
					if (cha >= 0) ya = sink_sample(ya, dither, ua0, ua1, p.sink.dither_mult, peak, clipped);
					if (chb >= 0) yb = sink_sample(yb, dither, ub0, ub1, p.sink.dither_mult, peak, clipped);
					if (dither) { ua0 = pm_mul(ua0, j0); ua1 = pm_mul(ua1, j1); ub0 = pm_mul(ub0, j0); ub1 = pm_mul(ub1, j1); }
					if (wpair) {
						char *dst = wout + (mo * p.C + cha) * bs;
						if (bs == 8) *reinterpret_cast<double2 *>(dst) = make_double2(ya, yb);
						else if (bs == 4) *reinterpret_cast<uint2 *>(dst) = make_uint2(pcm_to_word(ya, wf_sink), pcm_to_word(yb, wf_sink));
						else *reinterpret_cast<uint32_t *>(dst) = pcm_to_s16(ya) | (pcm_to_s16(yb) << 16);
					}
					else {
						if (cha >= 0) pcm_store(wout, p.sink.fmt, mo * p.C + cha, ya);
						if (chb >= 0) pcm_store(wout, p.sink.fmt, mo * p.C + chb, yb);
					}
				}
			};
			sink_set(v[0], j);
			sink_set(v[1], j + NTH);
			if (p.sink.stats) sink_stats_block(p.sink.stats, s, peak, clipped);
			continue;
		}
#pragma unroll
		for (int t = 0; t < VT; ++t) {
			const int jv = j + NTH * t;
#pragma unroll
			for (int m = 0; m < 16; ++m) {
				const int f = jv + P * m - first_n;
				if (f < f_lo || f >= f_hi) continue;
				cplx y = v[t][m];
				if (p.round_f32) { y.x = (double) (float) y.x; y.y = (double) (float) y.y; }
				if (p.ring_out) {
					if (p.ring_out_round_f32) { y.x = (double) (float) y.x; y.y = (double) (float) y.y; }
					if (chb < 0) y.y = 0.0;
					buf_stc(y, r_rout, ((rp0 + f) & omask) * 16);
				}
				else if (wide) buf_stc(y, r_out, ob + f * fb);
				else {
					if (cha >= 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sh_u32x2, y.x), r_out, ob + f * fb, 0, 0);
					if (chb >= 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sh_u32x2, y.y), r_out, ob2 + f * fb, 0, 0);
				}
			}
		}
	}
}

}  // namespace psh

template <int LOG2N, int BS> static void launch_short_t(const ShortParams &p, dim3 grid, hipStream_t st)
{
	grant_dynamic_lds(reinterpret_cast<const void *>(psh::conv_short<LOG2N, BS>), psh::ShCfg<LOG2N>::LDS);
	hipLaunchKernelGGL((psh::conv_short<LOG2N, BS>), grid, dim3(psh::ShCfg<LOG2N>::NTH), psh::ShCfg<LOG2N>::LDS, st, p);
}

void launch_conv_short(const ShortParams &p, hipStream_t st)
{
	if ((p.N != CONV_SHORT_N && p.N != CONV_SHORT_N2) || p.n_pairs < 1 || p.n_in < 1) return;
	const long n_blocks = (p.n_in + p.hop - 1) / p.hop;
	const long ranges = (n_blocks + p.blocks_per_wg - 1) / p.blocks_per_wg;
	const dim3 grid((unsigned) p.n_pairs, (unsigned) ranges);
	const int bs = (!p.slab || p.slab_fmt == PCM_DOUBLE) ? 8 : (p.slab_fmt == PCM_S16) ? 2 : 4;
	if (p.N == CONV_SHORT_N) { if (bs == 8) launch_short_t<13, 8>(p, grid, st); else if (bs == 4) launch_short_t<13, 4>(p, grid, st); else launch_short_t<13, 2>(p, grid, st); }
	else { if (bs == 8) launch_short_t<14, 8>(p, grid, st); else if (bs == 4) launch_short_t<14, 4>(p, grid, st); else launch_short_t<14, 2>(p, grid, st); }
}

}  // namespace dspamd
