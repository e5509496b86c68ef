// kernels_short.hip -- one-trip FFT convolution for SHORT filters behind long calls (round 5).
//
// What it replaces, per block, in the reference: fir_p's / fir's block transform, spectrum product and inverse (fir_p.c:64-103, fir.c:109-149) for
// filters of up to 4097 taps -- `hilbert -p 4095` (hilbert.c:28-92), crossover and correction FIRs, the FIRs `biquad -r` sections are designed into.
// The four-step convolver (kernels_fft.hip) sends every window through HBM three times whatever the filter's length; a window of 8192 points of a
// channel pair is 128 KB and fits one workgroup's LDS, so here a block is ONE read of the window (first_n frames of history + hop new ones) and ONE
// write of hop outputs: (N + hop) / hop units of 16 bytes per pair and frame -- 3 at 4095 taps, 2.3 at 1000 -- against 6.4 for three trips of a
// 65536-point transform.  BASELINE config 5's `hilbert -p 4095` stage was 22 of its 35 ms.
//
// What bounds it (round 5, BASELINE config 5's hilbert stage: 1024 pairs x 224 blocks, 44 GB in 15.6 ms = 2.8 TB/s, where the four-step path took 22 ms for
// 97 GB): not memory -- a block is two 8192-point transforms for 4096 outputs, about 7.5 us of fp64 issue and 6 us of LDS exchange traffic (six exchanges of
// 128 KB each way) per CU against 9 us of HBM time, in one workgroup whose eight waves meet at a barrier between every two of those phases.  The next window
// prefetched into a second register set (the filter row then read from L2 where it is used: 255 registers) measures 16.9 ms against 16.2 for the same
// code without it (profiles/r05_conv_short_prefetch_ab.txt, scripts/conv_short_prefetch_r05.patch): there is no idle memory time to fill.
//
// Workgroup = one channel pair (z = x_a + i x_b, exact: h is real), 512 threads x 16 points, walking the pair's blocks with the filter row in registers;
// radix 16 / 16 / 16 / 2 Stockham passes through the row buffer (the XOR-swizzled slots of conv_row: the same store and gather shapes).  One workgroup
// per CU (139 KB of LDS).  Same ring / slab / output conventions as K1 and K3 (fft_params.h: ShortParams).
#include <hip/hip_runtime.h>
#include <cstdint>
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

constexpr int SH_LOG2N = 13, SH_N = 1 << SH_LOG2N, SH_P = SH_N / 16, SH_THREADS = SH_P;
constexpr int SH_T256 = 272, SH_TLO = 68, SH_THI = SH_N / 64;
constexpr size_t SH_LDS = ((size_t) SH_N + SH_T256 + SH_TLO + SH_THI) * sizeof(cplx);

template <bool INV, class Tw>
__device__ __forceinline__ void short_fft(cplx (&v)[16], int j, cplx *lds, const RowMap &map, const Tw &tw)
{
	// (j made opaque per pass: the thread's twiddle and exchange addresses depend on j alone and would all be computed once and kept)
	asm volatile("" : "+v"(j));
	pass16<SH_LOG2N, 16, 1, INV, false>(v, j, lds, map, tw);
	lds_barrier();
	gather16<SH_LOG2N>(v, j, lds, map);
	lds_barrier();
	asm volatile("" : "+v"(j));
	pass16<SH_LOG2N, 16, 16, INV, false>(v, j, lds, map, tw);
	lds_barrier();
	gather16<SH_LOG2N>(v, j, lds, map);
	lds_barrier();
	asm volatile("" : "+v"(j));
	pass16<SH_LOG2N, 16, 256, INV, false>(v, j, lds, map, tw);
	lds_barrier();
	gather16<SH_LOG2N>(v, j, lds, map);
	asm volatile("" : "+v"(j));
	pass16<SH_LOG2N, 2, 4096, INV, true>(v, j, lds, map, tw);
}

// (every address is a buffer descriptor + a 32-bit offset: the host checks that rings, slabs and outputs stay below 2 GB per stream / pair -- sixteen
// 64-bit address pairs per direction do not fit beside the 64 registers of the window and the 64 of the filter row)
typedef unsigned int sh_u32x2 __attribute__((ext_vector_type(2)));
// BS: bytes per sample of the slab in direct mode (8: fp64; 4: s24 / s32 / float; 2: s16 -- read_buf_<fmt> of pcm_device.h in the loads)
template <int BS>
__global__ __launch_bounds__(SH_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_short(ShortParams p)
{
	constexpr int N = SH_N, P = SH_P;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	cplx *data = reinterpret_cast<cplx *>(smem_raw);
	cplx *t256 = data + N, *tlo = t256 + SH_T256, *thi = tlo + SH_TLO;
	int j = threadIdx.x;
	const long pair = blockIdx.x;
	const long n_blocks = (p.n_in + p.hop - 1) / p.hop;
	const long b0 = (long) blockIdx.y * p.blocks_per_wg, b1 = (b0 + p.blocks_per_wg < n_blocks) ? b0 + p.blocks_per_wg : n_blocks;
	if (b0 >= b1) return;
	if (j < 256) t256[twpad(j)] = TAB(p.tw)[j * (N / 256)];
	else if (j < 256 + 64) tlo[twpad(j - 256)] = TAB(p.tw)[j - 256];
	else if (j < 256 + 64 + SH_THI) thi[j - 320] = TAB(p.tw)[(j - 320) * 64];
	cplx h[16];
	if (!p.Hout) {
		const cplx *H = TAB(p.H) + (long) p.pair_h[pair] * N + j;
#pragma unroll
		for (int m = 0; m < 16; ++m) h[m] = H[P * m];
	}
	lds_barrier();                                                       // tables visible
	const TwRow<N> tw{ t256, tlo, thi };
	const RowMap map{ 0 };
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
		cplx v[16];
		asm volatile("" : "+v"(j));                                  // (addresses are recomputed per block: kept across blocks they are 40 registers nobody has)
#pragma unroll
		for (int m = 0; m < 16; ++m) {
			const int n = j + P * m;
			if (n >= valid) v[m] = mkc(0.0, 0.0);
			else if (n >= n_slab) {
				v[m] = slab_ld(so + n * fbi);
				if (n >= n_file) buf_stc(v[m], r_ring, ((r0 + n) & mask) * 16);
			}
			else v[m] = buf_ldc(r_ring, ((r0 + n) & mask) * 16, 0);
		}
		if (b > b0) lds_barrier();                                   // the previous block's last gather is done
		short_fft<false>(v, j, data, map, tw);
		if (p.Hout) {
			cplx *Ho = reinterpret_cast<cplx *>(p.Hout) + pair * N + j;
#pragma unroll
			for (int m = 0; m < 16; ++m) Ho[P * m] = mkc(v[m].x * p.h_scale, v[m].y * p.h_scale);
			return;
		}
#pragma unroll
		for (int m = 0; m < 16; ++m) { v[m] = cmul(v[m], h[m]); if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0); }
		lds_barrier();                                               // every gather of the forward transform is done
		short_fft<true>(v, j, data, map, tw);
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
			int m_first = 16;
#pragma unroll
			for (int m = 15; m >= 0; --m) { const int f = j + P * m - first_n; if (f >= f_lo && f < f_hi) m_first = m; }
			uint32_t ua0 = 0, ua1 = 0, ub0 = 0, ub1 = 0, j0 = 1, j1 = 1;
			if (dither && m_first < 16) {
				const long mo = mo0 + (j + P * m_first - first_n);
				const uint64_t na = (uint64_t) (p.sink.samples_before + mo * p.C + (cha >= 0 ? cha : 0)) + 1;
				const uint64_t nb = (uint64_t) (p.sink.samples_before + mo * p.C + (chb >= 0 ? chb : 0)) + 1;
				ua0 = pm_pow<0>(na); ua1 = pm_pow<1>(na);
				if (chb == cha + 1) { ub0 = pm_mul(ua0, PM_A0); ub1 = pm_mul(ua1, PM_A1); }
				else { ub0 = pm_pow<0>(nb); ub1 = pm_pow<1>(nb); }
				j0 = pm_pow<0>((uint64_t) P * p.C); j1 = pm_pow<1>((uint64_t) P * p.C);
			}
			double peak = 0.0;
			unsigned long long clipped = 0;
#pragma unroll
			for (int m = 0; m < 16; ++m) {
				const int f = j + P * m - first_n;
				if (f < f_lo || f >= f_hi) continue;
				const long mo = mo0 + f;
				double ya = v[m].x, yb = v[m].y;
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
			if (p.sink.stats) sink_stats_block(p.sink.stats, s, peak, clipped);
			continue;
		}
#pragma unroll
		for (int m = 0; m < 16; ++m) {
			const int f = j + P * m - first_n;
			if (f < f_lo || f >= f_hi) continue;
			cplx y = v[m];
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

}  // namespace psh

void launch_conv_short(const ShortParams &p, hipStream_t st)
{
	if (p.N != psh::SH_N || p.n_pairs < 1 || p.n_in < 1) return;
	const long n_blocks = (p.n_in + p.hop - 1) / p.hop;
	const long ranges = (n_blocks + p.blocks_per_wg - 1) / p.blocks_per_wg;
	const dim3 grid((unsigned) p.n_pairs, (unsigned) ranges), block(psh::SH_THREADS);
	const int bs = (!p.slab || p.slab_fmt == PCM_DOUBLE) ? 8 : (p.slab_fmt == PCM_S16) ? 2 : 4;
	if (bs == 8) { grant_dynamic_lds(reinterpret_cast<const void *>(psh::conv_short<8>), psh::SH_LDS); hipLaunchKernelGGL(psh::conv_short<8>, grid, block, psh::SH_LDS, st, p); }
	else if (bs == 4) { grant_dynamic_lds(reinterpret_cast<const void *>(psh::conv_short<4>), psh::SH_LDS); hipLaunchKernelGGL(psh::conv_short<4>, grid, block, psh::SH_LDS, st, p); }
	else { grant_dynamic_lds(reinterpret_cast<const void *>(psh::conv_short<2>), psh::SH_LDS); hipLaunchKernelGGL(psh::conv_short<2>, grid, block, psh::SH_LDS, st, p); }
}

}  // namespace dspamd
