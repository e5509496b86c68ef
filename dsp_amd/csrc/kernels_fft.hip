// kernels_fft.hip -- overlap-save FFT convolution for fir / fir_p / hilbert / zita-equivalent, hand-written for gfx950.
//
// Replaces the reference's FFTW-based paths
//   fir_effect_run                      fir.c:109-149      (one 2*len FFT per len samples)
//   fir_p_effect_run + fft_part_group_compute   fir_p.c:64-181   (non-uniform partitions + frequency-domain delay line)
// by their common mathematical content: y[n] = sum_k h[k] x[n-k] per channel, streamed.
//
// Design (docs/history.md section 4.2):
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

namespace p64 {
typedef double real;
#define FFT_F32 0
#include "fft_core.inc"
#undef FFT_F32
}  // namespace p64
using namespace p64;

// the float32-spectrum instance of the same kernels (kernels_fft32.hip)
namespace p32 {
void core_launch_conv_col(const ConvParams &p, bool inverse, int n_pairs, hipStream_t st);
void core_launch_row(const ConvParams &p, int mode, int n_pairs, hipStream_t st);
void core_launch_row_duo(const ConvParams &p, int n_pairs, hipStream_t st);
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
	// Rows of at least 1024 points belong to whole waves (P >= 64): the pair is wave-uniform -- said to the compiler (readfirstlane), so that the
	// stream, the channels and every base address derived from it live in scalar registers across the loop over the sub-blocks (as per-lane
	// values they cost the 4096-point instance 6 spilled registers until round 6)
	constexpr bool BUF = (P >= 64);
	const long pair_u = BUF ? (((long) __builtin_amdgcn_readfirstlane((int) ((active ? pair : 0) >> 32)) << 32) | (unsigned) __builtin_amdgcn_readfirstlane((int) (active ? pair : 0))) : 0;
	const long pair_e = BUF ? pair_u : (active ? pair : 0);
	const cplx *ring = p.ring + pair_e * p.ring_row_stride;
	const TwRow<NF> tw{ t256, tlo, thi };
	const RowMap map{ rw * Cfg::PITCH };
	// (a 64-bit division runs on the vector unit: its result is told to be uniform once more)
	const long s_v = pair_e / p.pairs_per_stream;
	const long s = BUF ? (((long) __builtin_amdgcn_readfirstlane((int) (s_v >> 32)) << 32) | (unsigned) __builtin_amdgcn_readfirstlane((int) s_v)) : s_v;
	const int qs = (int) (pair_e - s * p.pairs_per_stream);
	const int cha = p.pair_out_ch ? p.pair_out_ch[2 * qs] : -1, chb = p.pair_out_ch ? p.pair_out_ch[2 * qs + 1] : -1;
	// (64-bit products of uniform values still come out of the vector unit: the two base addresses are told to be uniform as well)
	auto uni64 = [](long v) { return BUF ? (((long) __builtin_amdgcn_readfirstlane((int) (v >> 32)) << 32) | (unsigned) __builtin_amdgcn_readfirstlane((int) v)) : v; };
	double *out = reinterpret_cast<double *>(uni64(reinterpret_cast<long>(p.out + (size_t) s * p.out_stride_frames * p.C)));
	const double *tail = p.tail ? reinterpret_cast<const double *>(uni64(reinterpret_cast<long>(p.tail + ((size_t) s * p.tail_stride_frames + p.tail_off) * p.C))) : nullptr;
	const bool wide = (chb == cha + 1) && ((cha & 1) == 0) && ((p.C & 1) == 0) && ((((size_t) out) & 15) == 0) && (!tail || (((size_t) tail) & 15) == 0);
	const size_t slot_stride = (size_t) p.n_pairs * NF;
	// With whole-wave rows every array is addressed as SGPR base (buffer descriptor) + ONE per-lane offset register + a scalar offset per
	// access.  With plain pointers the 16 strided accesses per array (4 KB and more apart: beyond the instruction's immediate) each hold a
	// 64-bit address pair: 87 scratch instructions at NF = 4096, two workgroups per CU.
	typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
	constexpr int RSRC_FLAGS = 0x00020000;                      // raw buffer, 32-bit offsets
	// (num_records: what the kernel is meant to touch from each base -- kparams.h rsrc_records; beyond it a load gives zeros, a store is dropped)
	const __amdgpu_buffer_rsrc_t r_fdl = __builtin_amdgcn_make_buffer_rsrc(p.fdl + (BUF ? pair_u * NF : 0), 0, rsrc_records((((long) p.P1 - 1) * p.n_pairs + 1) * NF * 16), RSRC_FLAGS);
	// the pair's filter (wave-uniform where the descriptors are used: rows of whole waves)
	const long h_off = (p.pair_h && active) ? (long) p.pair_h[pair_e] * p.P1 * NF : 0;
	const long h_off_u = BUF ? (((long) __builtin_amdgcn_readfirstlane((int) (h_off >> 32)) << 32) | (unsigned) __builtin_amdgcn_readfirstlane((int) h_off)) : 0;
	const __amdgpu_buffer_rsrc_t r_H = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx *>(p.Hf) + h_off_u, 0, rsrc_records((long) p.P1 * NF * 16), RSRC_FLAGS);
	const __amdgpu_buffer_rsrc_t r_ring = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx *>(p.ring) + (BUF ? pair_u * p.ring_row_stride : 0), 0, rsrc_records((p.ring_mask + 1) * 16), RSRC_FLAGS);
	// (32-bit byte offsets: the host only enters this regime when the delay line of a pair's slots and a ring row stay below 2 GB)
	const int jb = j * 16;
	const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out + (BUF ? (size_t) s * p.out_stride_frames * p.C : 0), 0, rsrc_records((long) p.n_sub * B * p.C * 8), RSRC_FLAGS);
	const __amdgpu_buffer_rsrc_t r_tail = __builtin_amdgcn_make_buffer_rsrc(
		const_cast<double *>(p.tail ? p.tail + ((BUF ? (size_t) s : 0) * p.tail_stride_frames + p.tail_off) * p.C : p.out), 0, p.tail ? rsrc_records((long) p.n_sub * B * p.C * 8) : 0, RSRC_FLAGS);
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
				const cplx *H = p.Hf + h_off + j;
#pragma unroll
				for (int m = 0; m < 16; ++m) v[m] = cmul(v[m], H[P * m]);
			}
			if (active) {
				for (int q = 1; q < p.P1; ++q) {
					const int sl = (slot + p.P1 - q) % p.P1;
					const cplx *X = line + (size_t) sl * slot_stride;
					const cplx *H = p.Hf + h_off + (size_t) q * NF + j;
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
		if (p.sink.on) {
			// the last kernel of a pipeline run in wire formats: dither, clip and convert in the stores (dsp.c:685-699).  A lane's
			// outputs are P frames apart, within a sub-block and from one to the next (B = 8 P): the generator values of its first
			// sample by modular exponentiation, the others by one multiplication with A^(P C)
			const bool dither = p.sink.dither_mult != 0.0;
			const int bs = (p.sink.fmt == PCM_DOUBLE) ? 8 : (p.sink.fmt == PCM_S16) ? 2 : 4;
			const WordFormat wf_sink = word_format(p.sink.fmt);
			char *wout = reinterpret_cast<char *>(p.out) + (size_t) s * p.out_stride_frames * p.C * bs;
			const bool wpair = (chb == cha + 1) && ((cha & 1) == 0) && ((p.C & 1) == 0) && ((((size_t) wout) & 15) == 0);
			const bool tpair = wpair && tail && ((((size_t) tail) & 15) == 0);
			double peak = 0.0;
			unsigned long long clipped = 0;
			if (active) {
				const long fr0 = (long) b * B + j;
				uint32_t ua0 = 0, ua1 = 0, ub0 = 0, ub1 = 0, j0 = 1, j1 = 1;
				if (dither) {
					const uint64_t na = (uint64_t) (p.sink.samples_before + fr0 * p.C + (cha >= 0 ? cha : 0)) + 1;
					ua0 = pm_pow<0>(na); ua1 = pm_pow<1>(na);
					if (chb == cha + 1) { ub0 = pm_mul(ua0, PM_A0); ub1 = pm_mul(ua1, PM_A1); }
					else { const uint64_t nb = (uint64_t) (p.sink.samples_before + fr0 * p.C + (chb >= 0 ? chb : 0)) + 1; ub0 = pm_pow<0>(nb); ub1 = pm_pow<1>(nb); }
					j0 = pm_pow<0>((uint64_t) P * p.C); j1 = pm_pow<1>((uint64_t) P * p.C);
				}
#pragma unroll
				for (int m = 8; m < 16; ++m) {
					const long f = fr0 + (long) P * (m - 8);
					double ya = v[m].x, yb = v[m].y;
					if (tpair) { const cplx t = *reinterpret_cast<const cplx *>(tail + f * p.C + cha); ya += t.x; yb += t.y; }
					else if (tail) { if (cha >= 0) ya += tail[f * p.C + cha]; if (chb >= 0) yb += tail[f * p.C + chb]; }
					if (p.round_f32) { ya = (double) (float) ya; yb = (double) (float) yb; }
					if (cha >= 0) ya = sink_sample(ya, dither, ua0, ua1, p.sink.dither_mult, peak, clipped);
					if (chb >= 0) yb = sink_sample(yb, dither, ub0, ub1, p.sink.dither_mult, peak, clipped);
					if (dither) { ua0 = pm_mul(ua0, j0); ua1 = pm_mul(ua1, j1); ub0 = pm_mul(ub0, j0); ub1 = pm_mul(ub1, j1); }
					if (wpair && bs != 1) {
						char *dst = wout + (f * p.C + cha) * bs;
						if (bs == 8) *reinterpret_cast<double2 *>(dst) = make_double2(ya, yb);
						else if (bs == 4) *reinterpret_cast<uint2 *>(dst) = make_uint2(pcm_to_word(ya, wf_sink), pcm_to_word(yb, wf_sink));
						else *reinterpret_cast<uint32_t *>(dst) = pcm_to_s16(ya) | (pcm_to_s16(yb) << 16);
					}
					else {
						if (cha >= 0) pcm_store(wout, p.sink.fmt, f * p.C + cha, ya);
						if (chb >= 0) pcm_store(wout, p.sink.fmt, f * p.C + chb, yb);
					}
				}
			}
			if (p.sink.stats) sink_stats_wave(p.sink.stats, s, active, peak, clipped);
			continue;
		}
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
					if (p.round_f32) { y.x = (double) (float) y.x; y.y = (double) (float) y.y; }
					__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), r_out, vo_out + so, 0, 0);
					continue;
				}
			}
			if (wide) {
				if (tail) { const cplx t = *reinterpret_cast<const cplx *>(tail + f * p.C + cha); y.x += t.x; y.y += t.y; }
				if (p.round_f32) { y.x = (double) (float) y.x; y.y = (double) (float) y.y; }
				*reinterpret_cast<cplx *>(out + f * p.C + cha) = y;
			}
			else {
				double ya = y.x + ((tail && cha >= 0) ? tail[f * p.C + cha] : 0.0), yb = y.y + ((tail && chb >= 0) ? tail[f * p.C + chb] : 0.0);
				if (p.round_f32) { ya = (double) (float) ya; yb = (double) (float) yb; }
				if (cha >= 0) out[f * p.C + cha] = ya;
				if (chb >= 0) out[f * p.C + chb] = yb;
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

static void pipe_grid(int groups, int n_pairs, int *ranges, int *per)
{
	// one workgroup per CU (LDS): the row groups times as many pair ranges as it takes to give every CU two workgroups in turn
	constexpr int wgs = 512;
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

// Which row kernel serves a plan -- decided from the plan alone (never from the number of pairs in a launch beyond "a few"):
//   rows of 2048 / 4096 points of one shared filter: the two-workgroup persistent kernel (conv_row_duo: plain convolution with the filter row in
//   registers; the 2x resampler's two branches with the rows from L2);
//   rows of 512 / 1024 points of one shared filter: the persistent kernel with the landing zone (conv_row_pipe);
//   everything else -- one filter per channel, the delay-line form (mode 3), filter preparation (mode 1), launches of a few pairs -- the one-shot kernel.
// (Round 6 removed the split-row kernel conv_row_big -- per-channel filters at 2048-point rows, 10 % ahead of the one-shot kernel there and nowhere
// the default of a BASELINE plan -- and the A/B switches DSP_AMD_ROW_PIPE / _ROW_BIG / _ROW_DUO / _ROW_DUO2: their measurements are in docs/history.md.)
static bool plan_is_shared(const ConvParams &p) { return p.nph <= 2 && p.shared_h; }

void launch_conv_row(const ConvParams &p, int mode, int n_pairs, hipStream_t st)
{
	if (mode == 3) { if (p.f32) p32::core_launch_row(p, 3, n_pairs, st); else core_launch_row(p, 3, n_pairs, st); return; }     // delay-line form: the generic row kernel
	const bool persistent = plan_is_shared(p) && n_pairs >= 8 && ((mode == 0 && p.nph == 1) || (mode == 2 && p.nph == 2));
	if (p.f32) {
		// the float32 instance: the persistent two-workgroup kernel for the long rows of a shared filter, else the one-shot kernel
		if (persistent && p.nph == 1 && p.log2N2 >= 11) p32::core_launch_row_duo(p, n_pairs, st);
		else p32::core_launch_row(p, mode, n_pairs, st);
		return;
	}
	if (persistent && p.log2N2 >= 11) { core_launch_row_duo(p, n_pairs, st); return; }
	if (persistent) {
		if (p.log2N2 == 9) { launch_row_pipe<9>(p, n_pairs, st); return; }
		if (p.log2N2 == 10) { launch_row_pipe<10>(p, n_pairs, st); return; }
	}
	core_launch_row(p, mode, n_pairs, st);
}

void launch_conv_col(const ConvParams &p, bool inverse, int n_pairs, hipStream_t st)
{
	if (p.f32) p32::core_launch_conv_col(p, inverse, n_pairs, st); else core_launch_conv_col(p, inverse, n_pairs, st);
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