// kernels_short.hip -- one-trip FFT convolution for SHORT filters behind long calls (round 5; round 6's form).
//
// What it replaces, per block, in the reference: fir_p's / fir's block transform, spectrum product and inverse (fir_p.c:64-103, fir.c:109-149) for
// filters of up to 8193 taps -- `hilbert -p 4095` (hilbert.c:28-92), crossover and correction FIRs, the FIRs `biquad -r` sections are designed into.
// The four-step convolver (kernels_fft.hip) sends every window through HBM three times whatever the filter's length; a window of 8192 or 16384 points of
// a channel pair fits one workgroup's registers (32 points per thread) with an exchange buffer of HALF its bytes in LDS, so here a block is ONE read of the
// window (first_n frames of history + hop new ones) and ONE write of hop outputs: (N + hop) / hop units of 16 bytes per pair and frame -- 2.3 at 4095 taps on
// the 16384-point window (3 on the 8192-point one), 2.3 at 1000 on the small one -- against 6.4 for three trips of a 65536-point transform.
//
// Workgroup = one channel pair (z = x_a + i x_b, exact: h is real), N / 32 threads x two sets of 16 points, walking the pair's blocks.  Transform: radix
// 32 / 16 / 16 (8192 points; 256 threads, 74 KB of LDS: two independent workgroups per CU) or 32 / 32 / 16 (16384 points; 512 threads, 148 KB: one) with TWO
// exchanges -- the thread's 32 points are the inputs of ITS radix-32 butterfly in every pass, so a radix-32 step is two 16-point transforms, constants and a
// radix-2 step in registers.  The exchanges carry real parts, then imaginary parts, through a buffer of N (+ N / 32) doubles.  The filter row comes from L2
// where it is used.  Same ring / slab / output conventions as K1 and K3 (fft_params.h: ShortParams).
//
// What bounds it (BASELINE config 5's hilbert stage: 1024 pairs of 917504 frames at 4095 taps, 35 GB on the large window): not memory.  Round 5 (one
// 512-thread workgroup per CU, 8192 points, radix 16 / 16 / 16 / 2, complex exchanges through 128 KB) 15.6 - 17.5 ms; wave-uniform descriptors 16.4 - 16.7;
// two point sets per thread and half-size exchanges (two workgroups per CU) 15.9; two exchanges instead of three 14.15; the 16384-point window (hop 12288
// instead of 4096: 14 / 13 of the work per point for three times the outputs) 11.5; then the instruction stream: 2500 of a wave's 5800 vector instructions
// per block were ADDRESSES (XOR-swizzled slots, two-level twiddle tables, three questions per loaded element) -- exchange slots pos + (pos >> 5), a
// twiddle table [r][k], last-pass twiddles as powers of one table entry, whole-window loads and whole-pair stores for blocks in the middle of a call:
// 9.4 ms = 3.7 TB/s (11.7 on the 8192-point window).  profiles/r06c_conv_short_*.txt, profiles/r06c_conv_short_config5_counters_*.json.
// A prefetched next window measured slower in round 5 (profiles/r05_conv_short_prefetch_ab.txt): there are no registers for it.  The filter row's bins asked
// for ahead of their products (two register sets, the first request behind the forward transform's last gather) measured the same 9.35 ms: not kept.
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
// j + N / 32: v[t][m] <-> position j + (N / 32) (t + 2 m), the 32 inputs of the thread's radix-32 butterfly in EVERY pass).  LDS: the exchange buffer (one
// half -- real or imaginary parts -- of the window at a time, one slot of padding per 32) and the middle pass's twiddles as a table [r][k] = W_(32 RB)^(r k),
// k < 32: the lanes of a wave read consecutive entries at an offset that is a constant of the instruction.
template <int LOG2N> struct ShCfg {
	static constexpr int N = 1 << LOG2N, P = N / 16, NTH = N / 32;
	static constexpr int RB = (LOG2N == 14) ? 32 : 16;                    // radix of the middle pass
	static constexpr int XCH = N + N / 32, TWB = RB * 32;
	static constexpr size_t LDS = (size_t) XCH * sizeof(double) + (size_t) TWB * sizeof(cplx);
};

// The exchanges carry the real and the imaginary parts one after the other through a buffer of doubles: twice the barriers and LDS instructions for the
// same bytes -- but half the LDS, so that two independent 8192-point workgroups fit a CU (one issues butterflies while the other waits), and a 16384-point
// window fits a CU at all.  Slots: an element is 8 bytes = 2 banks of 64; a ds_read / ds_write_b64 is served in two groups of 32 lanes, conflict-free when
// the 32 slots differ mod 32.  slot = pos + (pos >> 5) -- one slot of padding per 32 -- makes them for every access shape of the passes below AND leaves every
// address of a pass a per-thread base plus a constant of the instruction (the first form of this file XORed bits into the position: conflict-free too, but
// three integer instructions per access -- 2500 of a wave's 5800 vector instructions per block were addresses, profiles/r06c_conv_short_config5_counters_
// two_exchanges.json):
//   stores of the first pass      pos = 32 j + r               slot = 33 j + r
//   stores of the middle pass     pos = 32 RB a + k + 32 r     slot = 33 RB a + k + 33 r      (virtual thread 32 a + k)
//   gathers                       pos = j + (N / 32) m         slot = j + (j >> 5) + (N / 32 + N / 1024) m
// (simulated over every pass before it ran: scripts/conv_short_model.py)

// One radix-32 step on the thread's 32 points, u[2 m] = va[m], u[2 m + 1] = vb[m] (already multiplied by the pass's twiddles): 32 = 2 x 16 -- the two sets'
// 16-point transforms, w_32^k2 on the odd set's results (constants), a radix-2 step across the sets.  Output r ends up in va[r] (r < 16) or vb[r - 16].
template <bool INV>
__device__ __forceinline__ void sh_dft32(cplx (&va)[16], cplx (&vb)[16])
{
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

// The last pass of a set (radix 16, stride N / 16: butterfly = the virtual thread jv itself, results stay where the inputs were): twiddles w^r, r < 16,
// w = W_N^jv -- ONE table entry per set (read from the W_N table in device memory once per launch) and its powers by products at most four deep, in place of
// fifteen look-ups in a two-level table, their products and their index arithmetic.
template <bool INV>
__device__ __forceinline__ void sh_last16(cplx (&v)[16], const cplx w1)
{
	auto tw = [&](cplx &x, const cplx w) { x = INV ? cmulc(x, w) : cmul(x, w); };
	// w^(4 a + b) = w^(4 a) w^b: three small powers and one running multiple of four are alive at a time (the window's 128 registers leave room for little else)
	const cplx w2 = cmul(w1, w1), w3 = cmul(w2, w1);
	tw(v[1], w1); tw(v[2], w2); tw(v[3], w3);
	cplx w4a = cmul(w2, w2);
	const cplx w4 = w4a;
#pragma unroll
	for (int a = 1; a < 4; ++a) {
		tw(v[4 * a], w4a); tw(v[4 * a + 1], cmul(w4a, w1)); tw(v[4 * a + 2], cmul(w4a, w2)); tw(v[4 * a + 3], cmul(w4a, w3));
		if (a < 3) w4a = cmul(w4a, w4);
	}
	dft16<INV>(v);
}

// The N-point transform of a pair's window: radix 32 / 16 / 16 at 8192 points, 32 / 32 / 16 at 16384 -- TWO exchanges (until round 6's second half: radix
// 16 / 16 / 16 / 2 with three; a thread's 32 points are the same positions in every pass either way, so a radix-32 step costs no exchange of its own).
// Results in natural order at the positions the thread loaded from.  tk = the middle pass's twiddle table + (j & 31); wa, wb = W_N^j, W_N^(j + N / 32).
template <int LOG2N, bool INV>
__device__ __forceinline__ void short_fft2(cplx (&va)[16], cplx (&vb)[16], int j, double *lds, const cplx *tk, const cplx wa, const cplx wb)
{
	typedef ShCfg<LOG2N> Cfg;
	constexpr int P = Cfg::P, H = Cfg::NTH, PP = P + P / 32, HP = H + H / 32;
	double *const g = lds + (j + (j >> 5));                                 // gathers: set A at g[PP m], set B at g[HP + PP m]
	// sa, sb: where va[0], vb[0] go; dr: slots from one output of the butterfly to the next
	auto exchange = [&](double *const sa, double *const sb, auto dr_tag, bool last) {
		constexpr int DR = decltype(dr_tag)::value;
#pragma unroll
		for (int r = 0; r < 16; ++r) { sa[DR * r] = va[r].x; sb[DR * r] = vb[r].x; }
		lds_barrier();
		double xa[16], xb[16];
#pragma unroll
		for (int m = 0; m < 16; ++m) { xa[m] = g[PP * m]; xb[m] = g[HP + PP * m]; }
		lds_barrier();
#pragma unroll
		for (int r = 0; r < 16; ++r) { sa[DR * r] = va[r].y; sb[DR * r] = vb[r].y; }
		lds_barrier();
#pragma unroll
		for (int m = 0; m < 16; ++m) { va[m] = mkc(xa[m], g[PP * m]); vb[m] = mkc(xb[m], g[HP + PP * m]); }
		if (!last) lds_barrier();            // (behind the last exchange the caller's own barrier stands in front of the next store)
	};
	sh_dft32<INV>(va, vb);
	exchange(lds + 33 * j, lds + 33 * j + 16, std::integral_constant<int, 1>{}, false);
	auto tw = [&](cplx &x, const cplx w) { x = INV ? cmulc(x, w) : cmul(x, w); };
	if constexpr (LOG2N == 13) {
		// radix 16, stride 32, per set: virtual thread jv = 32 a + k, twiddles W_512^(r k)
#pragma unroll
		for (int r = 1; r < 16; ++r) { const cplx w = tk[32 * r]; tw(va[r], w); tw(vb[r], w); }      // (j and j + 256 have the same k)
		dft16<INV>(va);
		dft16<INV>(vb);
		double *const s = lds + (33 * 16) * (j >> 5) + (j & 31);
		exchange(s, s + (33 * 16) * (H / 32), std::integral_constant<int, 33>{}, true);
	}
	else {
		// radix 32, stride 32: thread j = 32 a + k, twiddles W_1024^(r k) on u[r] = (r even ? va : vb)[r / 2]
#pragma unroll
		for (int m = 0; m < 16; ++m) { if (m) tw(va[m], tk[32 * (2 * m)]); tw(vb[m], tk[32 * (2 * m + 1)]); }
		sh_dft32<INV>(va, vb);
		double *const s = lds + (33 * 32) * (j >> 5) + (j & 31);
		exchange(s, s + 33 * 16, std::integral_constant<int, 33>{}, true);
	}
	// (the powers are the same in every block: left to itself the compiler computes all thirty of them once per launch and keeps them -- in scratch memory)
	cplx w = wa;
	asm volatile("" : "+v"(w.x), "+v"(w.y));
	sh_last16<INV>(va, w);
	__builtin_amdgcn_sched_barrier(0);
	w = wb;
	asm volatile("" : "+v"(w.x), "+v"(w.y));
	sh_last16<INV>(vb, w);
}

// (every address is a buffer descriptor + a 32-bit offset: the host checks that rings, slabs and outputs stay below 2 GB per stream / pair -- sixteen
// 64-bit address pairs per direction do not fit beside the 64 registers of the window and the 64 of the filter row)
typedef unsigned int sh_u32x2 __attribute__((ext_vector_type(2)));
// BS: bytes per sample of the slab in direct mode (8: fp64; 4: s24 / s32 / float; 2: s16 -- read_buf_<fmt> of pcm_device.h in the loads)
// FDL: the uniformly partitioned form (windows from the rings only; an instance of its own: the plain instances do not carry its descriptors)
template <int LOG2N, int BS, bool FDL = false>
__global__ __launch_bounds__(ShCfg<LOG2N>::NTH) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_short(ShortParams p)
{
	static_assert(!FDL || BS == 8, "the delay-line form reads the rings");
	if constexpr (FDL) p.slab = nullptr;
	typedef ShCfg<LOG2N> Cfg;
	constexpr int N = Cfg::N, P = Cfg::P, NTH = Cfg::NTH, VT = 2;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	double *datad = reinterpret_cast<double *>(smem_raw);                // the exchange buffer: one half (real / imaginary parts) of the window at a time
	cplx *twb = reinterpret_cast<cplx *>(smem_raw + (size_t) Cfg::XCH * sizeof(double));
	int j = threadIdx.x;
	const long pair = blockIdx.x;
	const long n_blocks = (p.n_in + p.hop - 1) / p.hop;
	const long b0 = (long) blockIdx.y * p.blocks_per_wg, b1 = (b0 + p.blocks_per_wg < n_blocks) ? b0 + p.blocks_per_wg : n_blocks;
	if (b0 >= b1) return;
	for (int e = j; e < Cfg::TWB; e += NTH) twb[e] = TAB(p.tw)[(e >> 5) * (e & 31) * (N / Cfg::TWB)];      // [r][k] = W_(32 RB)^(r k)
	const cplx *const tk = twb + (j & 31);
	const cplx w_a = TAB(p.tw)[j], w_b = TAB(p.tw)[j + NTH];             // W_N^jv of the thread's two sets (the last pass's twiddles are their powers)
	const cplx *Hrow = p.Hout ? nullptr : TAB(p.H) + (long) p.pair_h[pair] * N;
	lds_barrier();                                                       // tables visible
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
	// (vo: the lane's byte offset, so: an offset every lane shares -- a scalar register of the instruction)
	auto slab_ld = [&](int vo, int so) -> cplx {
		if constexpr (BS == 8) return buf_ldc(r_slab, vo, so);
		else if constexpr (BS == 4) { const sh_u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(r_slab, vo, so, 0); return mkc(pcm_from_word(w.x, wf_slab), pcm_from_word(w.y, wf_slab)); }
		else { const uint32_t w = __builtin_amdgcn_raw_buffer_load_b32(r_slab, vo, so, 0); return mkc(pcm_from_s16(w & 0xffffu), pcm_from_s16(w >> 16)); }
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
		// A block in the middle of a long call reads its whole window from one place, with nothing to file and nothing to pad: the lane's offset once and
		// a scalar offset per element (the general form below asks three questions per element: 1600 instructions in front of the block's 32 loads).
		if (valid == N && p.slab && d_slab <= 0 && d_file >= N) {
#pragma unroll
			for (int t = 0; t < VT; ++t) {
				const int vo = so + (j + NTH * t) * fbi;
#pragma unroll
				for (int m = 0; m < 16; ++m) v[t][m] = slab_ld(vo, (P * m) * fbi);
			}
		}
		else if (valid == N && !p.slab && r0 + N <= mask + 1) {
#pragma unroll
			for (int t = 0; t < VT; ++t) {
				const int vo = (r0 + j + NTH * t) * 16;
#pragma unroll
				for (int m = 0; m < 16; ++m) v[t][m] = buf_ldc(r_ring, vo, (P * m) * 16);
			}
		}
		else {
#pragma unroll
			for (int t = 0; t < VT; ++t) {
				const int jv = j + NTH * t;
#pragma unroll
				for (int m = 0; m < 16; ++m) {
					const int n = jv + P * m;
					if (n >= valid) v[t][m] = mkc(0.0, 0.0);
					else if (n >= n_slab) {
						v[t][m] = slab_ld(so + n * fbi, 0);
						if (n >= n_file) buf_stc(v[t][m], r_ring, ((r0 + n) & mask) * 16);
					}
					else v[t][m] = buf_ldc(r_ring, ((r0 + n) & mask) * 16, 0);
				}
			}
		}
		if (b > b0) lds_barrier();                                   // the previous block's last gather is done
		short_fft2<LOG2N, false>(v[0], v[1], j, datad, tk, w_a, w_b);
		if (p.Hout) {
#pragma unroll
			for (int t = 0; t < VT; ++t) {
				cplx *Ho = reinterpret_cast<cplx *>(p.Hout) + pair * N + j + NTH * t;
#pragma unroll
				for (int m = 0; m < 16; ++m) Ho[P * m] = mkc(v[t][m].x * p.h_scale, v[t][m].y * p.h_scale);
			}
			return;
		}
		if constexpr (FDL) {
			// Uniformly partitioned form (the delay-line tail of the small-call regime, mid-size calls: conv.cpp): the window's spectrum becomes the newest
			// entry of the pair's frequency-domain delay line, Y = sum_q X[now - q] H_q over the fdl_P partitions of the pair's filter -- conv_row's mode 3
			// (fft_core.inc) without the two trips of W around it.  A thread owns the same 32 bins in every block and every launch, so an entry is only
			// ever re-read by the thread that wrote it (the blocks of a launch in order in ONE workgroup; the launches of a stream are ordered anyway).
			// One descriptor per slot and per partition: offsets stay inside one window whatever the batch's size.
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the entries this thread wrote in earlier blocks of this launch have left it)
			const int slot = __builtin_amdgcn_readfirstlane((p.fdl_slot + (int) (b - b0)) % p.fdl_P);   // (a division runs on the vector unit: its result is said to be uniform, or the descriptors below are per-lane values)
			const cplx *Hf = TAB(p.H) + (long) p.pair_h[pair] * p.fdl_P * N;
			cplx *line = reinterpret_cast<cplx *>(p.fdl) + pair * N;
			int vh = j * 16;
			asm volatile("" : "+v"(vh));
			// (FG bins per request group.  Four: 0.594 ms per run of the headline's seven-slot tail at 2048-frame calls (1024 pairs; K1 + K2 + K3: 0.62);
			// eight -- half the round trips, twice the requests in flight -- measured 0.617: the run is the bytes, 2.2 GB of windows and entries from HBM and
			// 1.8 GB of partition spectra from L2, not the chain of requests)
			constexpr int FG = 4;
			{
				const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(line + (long) slot * p.fdl_slot_stride, 0, rsrc_records((long) N * 16), 0x00020000);
				const __amdgpu_buffer_rsrc_t r_hq = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx *>(Hf), 0, rsrc_records((long) N * 16), 0x00020000);
#pragma unroll
				for (int t = 0; t < VT; ++t)
#pragma unroll
					for (int g = 0; g < 16 / FG; ++g) {
						cplx hh[FG];
#pragma unroll
						for (int m = 0; m < FG; ++m) hh[m] = buf_ldc(r_hq, vh + NTH * 16 * t, P * 16 * (FG * g + m));
#pragma unroll
						for (int m = 0; m < FG; ++m) {
							buf_stc(v[t][FG * g + m], r_x, vh + NTH * 16 * t + P * 16 * (FG * g + m));
							v[t][FG * g + m] = cmul(v[t][FG * g + m], hh[m]);
						}
						__builtin_amdgcn_sched_barrier(0);
					}
			}
			for (int q = 1; q < p.fdl_P; ++q) {
				const int sl = (slot >= q) ? slot - q : slot + p.fdl_P - q;
				const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(line + (long) sl * p.fdl_slot_stride, 0, rsrc_records((long) N * 16), 0x00020000);
				const __amdgpu_buffer_rsrc_t r_hq = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx *>(Hf + (long) q * N), 0, rsrc_records((long) N * 16), 0x00020000);
#pragma unroll
				for (int t = 0; t < VT; ++t)
#pragma unroll
					for (int g = 0; g < 16 / FG; ++g) {
						cplx x[FG], hh[FG];
#pragma unroll
						for (int m = 0; m < FG; ++m) { x[m] = buf_ldc(r_x, vh + NTH * 16 * t, P * 16 * (FG * g + m)); hh[m] = buf_ldc(r_hq, vh + NTH * 16 * t, P * 16 * (FG * g + m)); }
#pragma unroll
						for (int m = 0; m < FG; ++m) {
							cplx &a = v[t][FG * g + m];
							a.x = fma(x[m].x, hh[m].x, fma(-x[m].y, hh[m].y, a.x));
							a.y = fma(x[m].x, hh[m].y, fma(x[m].y, hh[m].x, a.y));
						}
						__builtin_amdgcn_sched_barrier(0);
					}
			}
		}
		else {
			// (no room for the filter row beside two sets of points: it comes from L2 where it is used -- the same 128 / 256 KB for every pair of a filter;
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
		short_fft2<LOG2N, true>(v[0], v[1], j, datad, tk, w_a, w_b);
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
		// A block in the middle of a long call: every window sample from first_n on is an output frame, whole pairs go to one place without a wrap --
		// one comparison and one addition per element (the general form below asks six questions per element).
		const bool all_frames = f_lo == 0 && f_hi == N - first_n && !p.round_f32;
		if (all_frames && p.ring_out && chb >= 0 && rp0 + f_hi <= omask + 1) {
			const bool rnd = p.ring_out_round_f32 != 0;
#pragma unroll
			for (int t = 0; t < VT; ++t) {
				const int f0 = j + NTH * t - first_n, vo = (rp0 + f0) * 16;
#pragma unroll
				for (int m = 0; m < 16; ++m) {
					if (f0 + P * m < 0) continue;
					cplx y = v[t][m];
					if (rnd) { y.x = (double) (float) y.x; y.y = (double) (float) y.y; }
					buf_stc(y, r_rout, vo + (P * m) * 16);
				}
			}
		}
		else if (all_frames && !p.ring_out && wide) {
#pragma unroll
			for (int t = 0; t < VT; ++t) {
				const int f0 = j + NTH * t - first_n, vo = ob + f0 * fb;
#pragma unroll
				for (int m = 0; m < 16; ++m) {
					if (f0 + P * m < 0) continue;
					buf_stc(v[t][m], r_out, vo + (P * m) * fb);
				}
			}
		}
		else {
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
}

}  // namespace psh

template <int LOG2N, int BS, bool FDL = false> static void launch_short_t(const ShortParams &p, dim3 grid, hipStream_t st)
{
	grant_dynamic_lds(reinterpret_cast<const void *>(psh::conv_short<LOG2N, BS, FDL>), psh::ShCfg<LOG2N>::LDS);
	hipLaunchKernelGGL((psh::conv_short<LOG2N, BS, FDL>), grid, dim3(psh::ShCfg<LOG2N>::NTH), psh::ShCfg<LOG2N>::LDS, st, p);
}

void launch_conv_short(const ShortParams &p_in, hipStream_t st)
{
	ShortParams p = p_in;
	if ((p.N != CONV_SHORT_N && p.N != CONV_SHORT_N2) || p.n_pairs < 1 || p.n_in < 1) return;
	const long n_blocks = (p.n_in + p.hop - 1) / p.hop;
	// (the delay-line form: every block of a pair in ONE workgroup, in order -- each reads what the ones before wrote)
	if (p.fdl_P > 0) p.blocks_per_wg = (int) n_blocks;
	const long ranges = (n_blocks + p.blocks_per_wg - 1) / p.blocks_per_wg;
	const dim3 grid((unsigned) p.n_pairs, (unsigned) ranges);
	if (p.fdl_P > 0 && !p.Hout) {
		if (p.N == CONV_SHORT_N) launch_short_t<13, 8, true>(p, grid, st); else launch_short_t<14, 8, true>(p, grid, st);
		return;
	}
	const int bs = (!p.slab || p.slab_fmt == PCM_DOUBLE) ? 8 : (p.slab_fmt == PCM_S16) ? 2 : 4;
	if (p.N == CONV_SHORT_N) { if (bs == 8) launch_short_t<13, 8>(p, grid, st); else if (bs == 4) launch_short_t<13, 4>(p, grid, st); else launch_short_t<13, 2>(p, grid, st); }
	else { if (bs == 8) launch_short_t<14, 8>(p, grid, st); else if (bs == 4) launch_short_t<14, 4>(p, grid, st); else launch_short_t<14, 2>(p, grid, st); }
}

}  // namespace dspamd
