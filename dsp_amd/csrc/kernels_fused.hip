// kernels_fused.hip -- the biquad cascade fused into the FFT convolver's first pass (round 4).
//
// What it replaces, per block, in the reference: the passes of biquad_effect_run (biquad.c:296-315, one per section, with
// biquad() of biquad.h:76-92 inside) followed by the input half of fir_p's block transform (fir_p.c:64-103) -- here ONE
// trip of the samples through HBM instead of two (cascade_rows writing the pair rings + conv_col_fwd reading them back).
//
// The four-step transform's first pass is the STRIDED one (the column of a window holds samples N2 frames apart), and a
// recurrence wants consecutive frames.  Both are served by giving a thread a whole ROW of the window: a recurrence thread owns the
// N2 consecutive frames of one row of one channel pair and walks along them 8 at a time, its section states in registers; the 256
// rows of a tile of 8 columns are then exactly 8 complete columns per pair, transformed on the spot.
// Rows are N2 frames apart in time, so the state a row starts from is not known to the kernel that walks all rows at once:
// a first kernel (fused_prepass) runs every row from ZERO state and keeps only the END state, cascade_chunk_carry
// (kernels_chunk.hip: x' = M x + e over the rows, M = A^N2 from the host in extended precision) turns those into the true state at
// every row start, and fused_col_fwd runs the recurrence again from those states -- the reference's own recurrence, sample by
// sample, no correction terms.  The recurrence costs its 5 fp64 operations per sample and section twice (cascade_rows spends
// 12 once: zero-state pass, row scan, zero-input correction) and the input is read twice; the cascade's output never goes to
// HBM except for the last `first_n` frames of the call, which are the next window's history (the pair rings, as before).
// HBM bytes per input sample of a step: 8 (prepass) + 8 + 8.5 N / hop (first pass) against 16 + 16 N / hop for the separate kernels.
//
// What it took to make fused_col_fwd faster than the two kernels it replaces (256 x 8 ch, ten sections, 983040-frame steps; cascade_rows
// + conv_col_fwd: 8.6 + 5.85 = 14.45 ms; fused_prepass + scan: 4.0 + 0.23):
//   10.65 ms  256 threads, a thread = a row of BOTH pairs (160 state registers, one wave per SIMD with the whole register file, 16-column
//             tiles needed 130 spilled registers, 8-column tiles none): 47 % VALU-busy, 37 % of the cycles in issue stalls nobody fills
//    8.95     512 threads, a thread = a row of ONE pair (80 state registers, two waves per SIMD), 8 points per thread through radix 8 / 8 / 4
//             passes; the section loop pinned sample by sample (the scheduler's own order kept 8 more registers per section alive twice
//             over -- the input-only products of a whole section first, the new states computed in the loop latch -- : 46 spills at ten
//             sections, none after)
//    8.5      the two pairs of a frame in adjacent lanes of the loads; section coefficients asked for a section ahead (scalar loads)
//    8.5      passes in place in the tile buffer, two buffers swapping roles: three barriers per tile instead of five (no change: the
//             kernel does not wait at barriers)
// What is left, by counter and by experiment (profiles/r04a_fzctr*_sq_counters.json, scripts/exp_fused.py dbg): without its loads and
// stores the kernel takes 6.8 ms (60 % VALU-busy: recurrence 2.9 ms of issue, transform 1.2, LDS traffic + barriers 2.0); the memory
// instructions add 1.7 ms whether their data comes from HBM or from the caches, whether they are issued in one burst or one per section,
// one tile ahead or two -- it is the CU's own load / store path, not bandwidth (33 GB in 8.5 ms) and not latency.
// Tried and dropped (round 4): the workgroup as two halves that run on their own, one per pair, with counters in LDS for barriers (gfx950 has one
// barrier per workgroup), so that one pair's transform would meet the other's recurrence on every SIMD -- each half must then load its own pair,
// 16 of a frame's 64 bytes per lane, and that alone took the kernel from 8.4 to 10.4 ms (with real barriers; 11.4 with the polled ones);
// the tile loads by LDS-DMA (global_load_lds_dwordx4 into a buffer laid out in the loading lanes' order: no staging registers, no ds_write pass,
// output bit-identical): 8.87 against 8.78 ms on the same box (commit 'experiment (kept in history)').
// Measured again at the end of round 4, same box, a / b / a / b: without the recurrence the kernel still takes 8.2-8.4 ms (8.3-8.8 with it), without
// the transform 6.8-7.1, without loads and stores 6.8-6.9 -- the sections' 2.7 ms of issue hide under the tile's chain of LDS and memory round
// trips.  Three ways to break that chain, none faster: the second pair's waves one barrier behind the first pair's (all 512 threads still load and
// stage; recurrence of one pair beside the passes of the other on every SIMD) 8.6-9.0 ms with the pairs in waves 0-3 / 4-7 and 10.1 with even / odd
// waves -- a wave alone in the recurrence stalls on its own dependent chains, two of them fill each other's gaps; the next tile asked for right
// behind the staging of this one (a whole tile's time in flight, loads issued before the stores they would otherwise wait behind) 8.7 against 8.4-8.6;
// the spectrum rows kept in registers and stored one per section of the NEXT tile's recurrence, beside its loads (239 registers) 8.43-8.50 against 8.35-8.59.
// And the overlap moved INSIDE the waves (fused_col_fwd_x, in the history of this file): the passes of tile it - 1 in the gaps between the sections of
// tile it -- LDS reads asked for in one gap, butterflies and writes in the next, behind the wait for the section's coefficients (scalar loads and LDS share
// a counter) --, the next tile by LDS-DMA into a landing buffer of its own (no staging registers; 242 registers, no spills once the passes' LDS addresses
// were base + constant instead of one register per slot), four barriers per tile, output bit-identical: 8.7-9.0 ms against 8.5-8.8, VALU-busy 50 % in
// both (profiles/r04_fzx_sq_counters.json).  The clock is not the limit either: 2.04 GHz in this kernel, 2.04 in K2, 2.31 in K3, 1.79 in a pure fp64
// VALU kernel, 2.40 idle-ish (profiles/r04_kernel_clocks.json, scripts/exp_clock_step.sh).
//
// Layout of fused_col_fwd: workgroup = (stream, group of 2 channel pairs, row segment), 512 threads, one workgroup per CU (152 KB of
// LDS: two tile buffers [2 pairs][256 rows] of pitch 9).  The two groups of a stream are dispatched 8 workgroup ids apart (same XCD,
// same L2): each owns 32 of the 64 bytes of every frame (PMC: the slab is fetched once, profiles/r04a_traffic.json).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdint>
#include "kparams.h"
#include "fft_params.h"
#include "pcm_device.h"

namespace dspamd {
namespace pfz {
typedef double real;
#define FFT_F32 0
#define FFT_CORE_NO_LAUNCHERS 1
#include "fft_core.inc"
#undef FFT_CORE_NO_LAUNCHERS
#undef FFT_F32

// NSEC sections on NF consecutive frames of one channel pair (x[i].x / .y = the two channels), states carried in (m0, m1):
// biquad.h:76-92 -- r = c0 s + m0;  m0 = m1 + c1 s - c3 r;  m1 = c2 s - c4 r.  Coefficients are wave-uniform (scalar loads).
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
struct SecCoef { double c0, c1, c2, c3, c4; };
__device__ __forceinline__ SecCoef load_coef(const double *__restrict__ sec, int k) { return { sec[6 * k], sec[6 * k + 1], sec[6 * k + 2], sec[6 * k + 3], sec[6 * k + 4] }; }
// `between(k)` is called in front of section k: a place to put one memory instruction each.
// `cf` holds the coefficients of section 0 on entry and again on exit: every section asks for the next one's (scalar loads) before its
// own samples, the last one for section 0's -- asked for where they are used, the scalar cache's latency stood in front of every
// section (9 waits per 16-frame step of the prepass: a quarter of its time).
template <int NSEC, int NF, class Hook = NoHook>
__device__ __forceinline__ void run_sections(cplx (&x)[NF], double2 (&m0)[NSEC], double2 (&m1)[NSEC], const double *__restrict__ sec, SecCoef &cf, const Hook &between = Hook())
{
#pragma unroll
	for (int k = 0; k < NSEC; ++k) {
		between(k);
		const SecCoef cur = cf;
		cf = load_coef(sec, k + 1 < NSEC ? k + 1 : 0);
		__builtin_amdgcn_sched_barrier(0);
		const double c0 = cur.c0, c1 = cur.c1, c2 = cur.c2, c3 = cur.c3, c4 = cur.c4;
		double a0x = m0[k].x, a0y = m0[k].y, a1x = m1[k].x, a1y = m1[k].y;
#pragma unroll
		for (int i = 0; i < NF; ++i) {
			// (one sample of both channels at a time, in this order: left to itself the scheduler computes the input-only products of all
			// NF samples of a section first -- two registers each -- and a ten-section instance no longer fits 256 registers)
			const double sa = x[i].x, sb = x[i].y;
			const double ra = fma(c0, sa, a0x), rb = fma(c0, sb, a0y);
			const double ta = fma(c1, sa, a1x), tb = fma(c1, sb, a1y);
			const double ua = c2 * sa, ub = c2 * sb;
			a0x = fma(-c3, ra, ta); a0y = fma(-c3, rb, tb);
			a1x = fma(-c4, ra, ua); a1y = fma(-c4, rb, ub);
			x[i].x = ra; x[i].y = rb;
			__builtin_amdgcn_sched_barrier(0);
		}
		// (the new states are wanted HERE: left alone they are computed in the loop latch, from the section's last input and output kept
		// alive until then -- eight more registers per section)
		asm volatile("" : "+v"(a0x), "+v"(a0y), "+v"(a1x), "+v"(a1y));
		m0[k].x = a0x; m0[k].y = a0y; m1[k].x = a1x; m1[k].y = a1y;
	}
}

// ---- pass 0: the end state of every chunk run from zero state.  grid (ceil(K pps / 256), S): thread = (chunk c, pair q) of
// stream blockIdx.y, pair-fastest so that the lanes of one chunk read whole frames; every thread walks its len frames, 16 at a
// time, the next 16 already on their way.
#ifndef FZ_PRE_WAVES
#define FZ_PRE_WAVES 2
#endif
// PP: one section table per channel pair (f.sec_stride doubles apart): the lanes of a wave hold different pairs, so the coefficients are per-lane
// loads (L1-resident tables) instead of scalar ones, and a step is 8 frames (their registers).
template <int NSEC, bool PP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FZ_PRE_WAVES, FZ_PRE_WAVES)))
void fused_prepass(FuseParams f, const double *__restrict__ sec_all, long N2, int pps)
{
	const long id = (long) blockIdx.x * 256 + threadIdx.x;
	const long s = blockIdx.y;
	const bool live = id < f.K * pps;
	const int q = (int) (id % pps);
	const double *__restrict__ sec = PP ? sec_all + (size_t) q * f.sec_stride : sec_all;
	const long c = live ? id / pps : 0;
	const long row = c / f.seg;
	const int sg = (int) (c - row * f.seg);
	const long frame0 = row * N2 + (long) sg * f.len;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(f.in) + (size_t) s * f.in_stride_frames * f.C, 0, rsrc_records(f.K * f.len * f.C * (long) sizeof(double)), 0x00020000);   // (the call's frames: K chunks of len)
	const int fb = f.C * (int) sizeof(double);                      // bytes per frame
	int vo = (int) ((frame0 * f.C + 2 * q) * (long) sizeof(double));
	double2 m0[NSEC], m1[NSEC];
#pragma unroll
	for (int k = 0; k < NSEC; ++k) { m0[k] = make_double2(0.0, 0.0); m1[k] = make_double2(0.0, 0.0); }
#ifdef FZ_PRE_NF
	constexpr int NF = FZ_PRE_NF;
#else
	constexpr int NF = (NSEC > 10 || PP) ? 8 : 16;     // frames per step (twelve sections with 16: spills)
#endif
	cplx x[NF], nx[NF];
#pragma unroll
	for (int i = 0; i < NF; ++i) nx[i] = buf_ldc(rs, vo, i * fb);
	SecCoef cf = load_coef(sec, 0);
	for (long i0 = 0; i0 < f.len; i0 += NF) {
#pragma unroll
		for (int i = 0; i < NF; ++i) x[i] = nx[i];
		if (i0 + NF < f.len) {
			vo += NF * fb;
#pragma unroll
			for (int i = 0; i < NF; ++i) nx[i] = buf_ldc(rs, vo, i * fb);
		}
		run_sections<NSEC, NF>(x, m0, m1, sec, cf);
	}
	if (!live) return;
	const int D = 2 * f.n_ops;
	double *dst = f.cstate + (((size_t) s * f.K + c) * f.C + 2 * q) * D;
#pragma unroll
	for (int k = 0; k < NSEC; ++k) {
		const int op = f.sec_op[k];
		if (op < 0) continue;
		dst[2 * op] = m0[k].x; dst[2 * op + 1] = m1[k].x;
		dst[D + 2 * op] = m0[k].y; dst[D + 2 * op + 1] = m1[k].y;
	}
}

// ---- pass 0 on the matrix cores.  The end state of a chunk run from zero state is LINEAR in its samples: e = sum_t A^(len-1-t) B x[t]
// with (A, B) the state-space form of the whole cascade -- a product E[state][column] = G[state][t] X[t][column] over the chunk's frames
// with the same G for every (chunk, channel) column: 2 n_sec multiply-adds per sample (padded to 32 rows) where the recurrence
// itself costs 5 n_sec instructions.  G (host, extended precision: CascadeStage::fuse_gtable) is the state the cascade is left in
// len - 1 - t samples after a unit impulse.  v_mfma_f64_16x16x4_f64: A = G^T rows from L2 ([t][32], 1 MB, the same for every wave),
// B = four consecutive frames of 16 columns = (4 chunks) x (4 channel pairs), a lane's 16-byte load feeding two products (the pair's
// two channels); a wave owns 8 chunks (two such column groups) of one stream: the fragments of G are used four times.
// 8 channels per stream only (the launcher checks); anything else takes the recurrence above.
typedef double fz_v4d __attribute__((ext_vector_type(4)));

// one frame's channel pair (2 cp, 2 cp + 1) of a slab of samples of a wire format as fp64 (read_buf_<fmt>, pcm_device.h); bs = bytes per sample
// (BS is a template parameter of the kernels: chosen at run time, the width of the load was a scalar branch in front of every load)
template <int BS> __device__ __forceinline__ double2 fz_wire_pair(const char *frame_pair, const WordFormat &wf)
{
	if constexpr (BS == 4) { const uint2 w = *reinterpret_cast<const uint2 *>(frame_pair); return make_double2(pcm_from_word(w.x, wf), pcm_from_word(w.y, wf)); }
	else { const uint32_t w = *reinterpret_cast<const uint32_t *>(frame_pair); return make_double2(pcm_from_s16(w & 0xffffu), pcm_from_s16(w >> 16)); }
}

template <int DT, int BS = 8>            // DT: 16-row tiles of states: 1 (up to 8 sections) or 2; BS: bytes per sample of the slab (8: fp64, 2: s16, 4: s24 / s32 / float)
__global__ __launch_bounds__(256) void fused_prepass_mm(FuseParams f, const double *__restrict__ Gt, long N2, int n_state)
{
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const long s = blockIdx.y;
	const long c0 = ((long) blockIdx.x * 4 + w) * 8;          // this wave's first chunk
	if (c0 >= f.K) return;
	// (K need not be a multiple of 8 -- 239 rows behind 17 rows of history: the columns past the last chunk read the last chunk again and store nothing)
	const int k = lane >> 4, jc = lane & 15, cp = jc & 3;
	// a lane's element of step t0: frame t0 + k of chunk (jc >> 2) of the column group, channel pair cp -- 16 bytes of an fp64 slab, 4 or 8 of a wire format
	constexpr int bs = BS;
	const WordFormat wf = word_format(f.in_fmt);
	const char *xb[2];
#pragma unroll
	for (int g = 0; g < 2; ++g) {
		const long c = (c0 + 4 * g + (jc >> 2) < f.K) ? c0 + 4 * g + (jc >> 2) : f.K - 1;
		const long row = c / f.seg, frame0 = row * N2 + (c - row * f.seg) * f.len;
		xb[g] = reinterpret_cast<const char *>(f.in) + (((size_t) s * f.in_stride_frames + frame0 + k) * 8 + 2 * cp) * bs;
	}
	const long fstep = 8L * bs;                                   // bytes per frame (8 channels)
	auto xload = [&](int g, long t) -> double2 {
		if constexpr (BS != 8) return fz_wire_pair<BS>(xb[g] + t * fstep, wf);
		// (non-temporal: the slab is read once more, 16 GB later, by the first pass -- nothing of it survives in a cache anyway; round 5, a / b / a on one
		// box: 2.78 -> 2.59 ms.  The same hint on the first pass's own slab loads and on K3's W loads / slab stores measures nothing: profiles/r05_*)
		else { typedef double fz_d2 __attribute__((ext_vector_type(2))); const fz_d2 v = __builtin_nontemporal_load(reinterpret_cast<const fz_d2 *>(xb[g] + t * fstep)); return make_double2(v.x, v.y); }
	};
	const double *ga = Gt + (size_t) k * 32 + jc;
	fz_v4d acc[2][2][DT];
#pragma unroll
	for (int g = 0; g < 2; ++g)
#pragma unroll
		for (int eo = 0; eo < 2; ++eo)
#pragma unroll
			for (int dt = 0; dt < DT; ++dt) acc[g][eo][dt] = (fz_v4d) { 0.0, 0.0, 0.0, 0.0 };
	double2 x0[2], x1[2];
	double a0[DT], a1[DT];
#pragma unroll
	for (int g = 0; g < 2; ++g) x0[g] = xload(g, 0);
#pragma unroll
	for (int dt = 0; dt < DT; ++dt) a0[dt] = ga[16 * dt];
	// two steps of four frames per iteration, the next step's fragments asked for before this step's products (len is a multiple of 8)
	for (long t0 = 0; t0 < f.len; t0 += 8) {
#pragma unroll
		for (int g = 0; g < 2; ++g) x1[g] = xload(g, t0 + 4);
#pragma unroll
		for (int dt = 0; dt < DT; ++dt) a1[dt] = ga[(size_t) (t0 + 4) * 32 + 16 * dt];
#pragma unroll
		for (int g = 0; g < 2; ++g)
#pragma unroll
			for (int dt = 0; dt < DT; ++dt) {
				acc[g][0][dt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[dt], x0[g].x, acc[g][0][dt], 0, 0, 0);
				acc[g][1][dt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[dt], x0[g].y, acc[g][1][dt], 0, 0, 0);
			}
		const long tn = (t0 + 8 < f.len) ? t0 + 8 : t0;                               // (the last iteration re-reads: the body stays uniform)
#pragma unroll
		for (int g = 0; g < 2; ++g) x0[g] = xload(g, tn);
#pragma unroll
		for (int dt = 0; dt < DT; ++dt) a0[dt] = ga[(size_t) tn * 32 + 16 * dt];
#pragma unroll
		for (int g = 0; g < 2; ++g)
#pragma unroll
			for (int dt = 0; dt < DT; ++dt) {
				acc[g][0][dt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[dt], x1[g].x, acc[g][0][dt], 0, 0, 0);
				acc[g][1][dt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[dt], x1[g].y, acc[g][1][dt], 0, 0, 0);
			}
	}
	// results: row (state) = (lane >> 4) + 4 reg + 16 dt, column = lane & 15 = (chunk, channel pair); state 2 kk + b -> (m0, m1)[b] of section kk
	const int D = 2 * f.n_ops;
#pragma unroll
	for (int g = 0; g < 2; ++g) {
		const long c = c0 + 4 * g + (jc >> 2);
		if (c >= f.K) continue;
#pragma unroll
		for (int eo = 0; eo < 2; ++eo) {
			double *dst = f.cstate + (((size_t) s * f.K + c) * f.C + 2 * cp + eo) * D;
#pragma unroll
			for (int dt = 0; dt < DT; ++dt)
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const int d = k + 4 * r + 16 * dt;
					if (d < n_state) dst[2 * f.sec_op[d >> 1] + (d & 1)] = acc[g][eo][dt][r];
				}
		}
	}
}

// ---- pass 1: K1 with the cascade in front of its column transforms (see the head of the file).
// grid: S x groups x seg workgroups in the order of fz_block (the groups of a stream on one XCD), 512 threads = 8 waves, two per SIMD.
// HR = the window rows that are history (the pair rings), the others are new frames (the slab): 16 or 32 as compile-time constants (the headline's
// instances), or 0 = f.hist_rows at run time, any count from 1 to 32 (e.g. 17 rows behind the 66119 taps of fir_p merged into a 2x resampler).
// A tile = 8 columns of the group's two pairs.  Thread roles:
//   loading      (pair lq, column lt, lj): the rows lj + 32 m of a column, the two pairs of a frame (32 contiguous bytes) in adjacent lanes
//   recurrence   (pair rq, row rr): 8 consecutive frames of its row through the sections, states in registers (80)
//   transform    (pair q, column t, j): the 8 points n1 = j + 32 m of a column -- three passes of radix 8, 8, 4, in place in the tile buffer
constexpr int FZ_TW = 8, FZ_PITCH = 9, FZ_PT = 8, FZ_P = 256 / FZ_PT;
// one pair's rows in a buffer; = 8 mod 16: the 16 lanes of a read group of the two exchanges (two pairs x two rows 32 apart) land in 16 different
// bank quads (the same group reading the cascade's output -- rows one apart -- collides in 7 of them, the staging writes two ways: the cheaper side)
constexpr int FZ_QS = 256 * FZ_PITCH + 8;
constexpr size_t FZ_LDS = ((size_t) 4 * FZ_QS + 256) * sizeof(cplx);

__device__ __forceinline__ void fz_block(int n_streams, int n_gs, int &s, int &gs)
{
	const int id = blockIdx.x;
	if ((n_streams & 7) == 0) {
		const int per = 8 * n_gs, blk = id / per, r = id % per;
		s = blk * 8 + (r & 7);
		gs = r >> 3;
	}
	else { s = id / n_gs; gs = id % n_gs; }
}

// One Stockham pass on the PT register-resident points of a thread (pass16 of fft_core.inc for another number of points):
// v[m] <-> position j + P m, P = N / PT; PT / R butterflies b = j + P q of radix R whose inputs b + (N / R) r are v[q + (PT / R) r].
template <int PT, int LOG2N, int R, int NS, bool LAST, class Map, class Tw>
__device__ __forceinline__ void fz_pass(cplx (&v)[PT], int j, cplx *lds, const Map &map, const Tw &tw)
{
	constexpr int N = 1 << LOG2N, P = N / PT, Q = PT / R;
#pragma unroll
	for (int qq = 0; qq < Q; ++qq) {
		const int b = j + P * qq;
		const int k = b & (NS - 1);
		cplx u[R];
#pragma unroll
		for (int r = 0; r < R; ++r) u[r] = v[qq + Q * r];
		if constexpr (NS > 1) {
#pragma unroll
			for (int r = 1; r < R; ++r) u[r] = cmul(u[r], tw.template get<R * NS>(r * k));
		}
		dftR<R, false>(u);
		if constexpr (LAST) {
#pragma unroll
			for (int r = 0; r < R; ++r) v[qq + Q * r] = u[r];
		}
		else {
			const int j0 = (b - k) * R + k;
#pragma unroll
			for (int r = 0; r < R; ++r) map.store(lds, j0 + NS * r, u[r]);
		}
	}
}
template <int PT, class Map> __device__ __forceinline__ void gather_n(cplx (&v)[PT], int j, const cplx *lds, const Map &map)
{
#pragma unroll
	for (int m = 0; m < PT; ++m) map.load(lds, j + (256 / PT) * m, v[m]);
}
struct FzTw { const cplx *t; template <int M> __device__ __forceinline__ cplx get(int e) const { return t[e * (256 / M)]; } };   // W_256 table
// The column's 256 points live in the 256 slots (pair, row, column) of the tile buffer the whole time: a thread reads the rows j + 32 m
// the cascade left there, and every pass writes its 8 outputs to the 8 slots the thread has just read (so only other threads'
// READS need a barrier, never a free buffer).  Where a position of the Stockham sequence lives after that:
//   after pass 1 (thread j wrote position 8 j + r to row j + 32 r):                  position p at row (p >> 3) + 32 (p & 7)
//   after pass 2 (thread j, k = j & 7, wrote position 64 (j >> 3) + 8 r + k to the row it read its input r from,
//                 (j >> 3) + 4 r + 32 k):                                            position p at row (p >> 6) + 4 ((p >> 3) & 7) + 32 (p & 7)
struct FzMap1 {
	int base;            // pair * FZ_QS + column
	__device__ __forceinline__ static int row(int pos) { return (pos >> 3) + 32 * (pos & 7); }
	__device__ __forceinline__ void store(cplx *lds, int pos, cplx v) const { lds[base + row(pos) * FZ_PITCH] = v; }
	__device__ __forceinline__ void load(const cplx *lds, int pos, cplx &v) const { v = lds[base + row(pos) * FZ_PITCH]; }
};
struct FzMap2 {
	int base;
	__device__ __forceinline__ static int row(int pos) { return (pos >> 6) + 4 * ((pos >> 3) & 7) + 32 * (pos & 7); }
	__device__ __forceinline__ void store(cplx *lds, int pos, cplx v) const { lds[base + row(pos) * FZ_PITCH] = v; }
	__device__ __forceinline__ void load(const cplx *lds, int pos, cplx &v) const { v = lds[base + row(pos) * FZ_PITCH]; }
};

// the inter-pass twiddle w_N^(n2 k1) for the rows k1 = j + 32 m of a thread, from s = w_N^(16 n2) and a = w_N^(n2 (j & 15)) (tw_col's rows)
__device__ __forceinline__ void fz_twiddle(cplx s16, cplx a, int j, cplx (&v)[FZ_PT])
{
	if (j >= 16) a = cmul(a, s16);
	const cplx s1 = cmul(s16, s16), s2 = cmul(s1, s1), s3 = cmul(s2, s1), s4 = cmul(s2, s2);      // w^(32 n2) and its powers
	const cplx a4 = cmul(a, s4);
	v[0] = cmul(v[0], a); v[1] = cmul(v[1], cmul(a, s1)); v[2] = cmul(v[2], cmul(a, s2)); v[3] = cmul(v[3], cmul(a, s3));
	v[4] = cmul(v[4], a4); v[5] = cmul(v[5], cmul(a4, s1)); v[6] = cmul(v[6], cmul(a4, s2)); v[7] = cmul(v[7], cmul(a4, s3));
}

template <int NSEC, int HR, int BS = 8>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fused_col_fwd(ConvParams p, FuseParams f, const double *__restrict__ sec_all)
{
	const double *__restrict__ sec = sec_all;
	constexpr int N1 = 256, TW = FZ_TW, PT = FZ_PT, P = FZ_P;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	cplx *buf0 = reinterpret_cast<cplx *>(smem_raw);     // two tile buffers [2 pairs][256 rows] of pitch 9 that swap roles from tile to tile
	cplx *twt = buf0 + 4 * FZ_QS;
	const int tid = threadIdx.x;
	// three roles of a thread, three ways to number the tile's elements:
	const int lq = tid & 1, lt = (tid >> 1) & (TW - 1), lj = tid >> 4;   // loading: rows lj + 32 m of column lt of pair lq -- the two pairs of a frame (32 contiguous bytes) in adjacent lanes
	const int rr = tid & (N1 - 1), rq = tid >> 8;                        // recurrence: row rr of pair rq (a wave = 64 consecutive rows of one pair)
	const int t = tid & (TW - 1), q = (tid >> 3) & 1, j = tid >> 4;      // transform: points n1 = j + 32 m of column t of pair q (128-byte runs of W per 8 lanes)
	const int groups = p.pairs_per_stream >> 1;
	int s, gs;
	fz_block(f.n_streams, groups * f.seg, s, gs);
	const int grp = gs % groups, sg = gs / groups;
	if (tid < N1) twt[tid] = TAB(p.tw_n1)[tid];
	const long N2 = p.N2;
	const int tiles = (int) (N2 / TW / f.seg);
	const long col0 = (long) sg * tiles * TW;        // first column of this workgroup's segment
	const long pair0 = (long) s * p.pairs_per_stream + 2 * grp;     // the group's first pair
	// the slab through a buffer descriptor: element (row lj + 32 m >= HR, column lt) of pair lq at vs + (column + (32 m - HR) N2) frame bytes
	// (BS != 8: the slab holds samples of f.in_fmt -- s16, or s24 / s32 / float --, a pair of a frame is one 4- or 8-byte load, converted as it arrives)
	constexpr int bs = BS;
	const WordFormat wf = word_format(f.in_fmt);
	const int fb = f.C * bs;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(f.in)) + (size_t) s * f.in_stride_frames * f.C * bs, 0, rsrc_records(f.K * f.len * f.C * (long) bs), 0x00020000);   // (the call's frames: K chunks of len)
	const int vs = (int) ((((long) lj * N2 + lt) * f.C + 4 * grp + 2 * lq) * (long) bs);
	const int hr = HR > 0 ? HR : f.hist_rows;        // (a constant in the HR > 0 instances)
	const int vs0 = vs - (int) (hr * N2 * fb);       // row lj itself (looked at when hr <= lj: hr < 32)
	auto slab_ld = [&](int vo, int so) -> cplx {
		if constexpr (BS == 8) return buf_ldc(rs, vo, so);
		else if constexpr (BS == 4) {
			typedef unsigned int fz_u32x2 __attribute__((ext_vector_type(2)));
			const fz_u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0);
			return mkc(pcm_from_word(w.x, wf), pcm_from_word(w.y, wf));
		}
		else {
			const uint32_t w = __builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0);
			return mkc(pcm_from_s16(w & 0xffffu), pcm_from_s16(w >> 16));
		}
	};
	const bool hist_row = lj < hr;                   // row lj (m = 0) is history: the pair rings (wave-uniform when hr is a multiple of 4)
	// W of the group's two pairs through one descriptor
	const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(WBUF(p.W) + (pair0 - p.pair0) * p.w_stride, 0, rsrc_records((p.w_stride + p.N) * (long) sizeof(cplx)), 0x00020000);   // (the group's two pairs)
	const int vw = (int) (((long) q * p.w_stride + (long) j * N2 + t) * (long) sizeof(cplx));
	const int w_step = (int) (32 * N2 * (long) sizeof(cplx));
	const double2 *ring0 = p.ring + pair0 * p.ring_row_stride;
	const double2 *ringl = ring0 + lq * p.ring_row_stride;
	// so = the tile's first column in bytes of a slab row, wave-uniform BY CONSTRUCTION (readfirstlane): left to the compiler's own analysis it ended
	// up in a VGPR -- a 32-bit VALU multiply per load and a waterfall loop around every buffer load to get the scalar offset back
	auto tile_so = [&](int it) { return __builtin_amdgcn_readfirstlane((int) ((col0 + (long) it * TW) * fb)); };
	auto fetch1 = [&](int it, int so, int m, cplx (&d)[PT]) {    // (m is a compile-time constant at every call)
		const bool from_ring = (32 * m + 32 <= hr) || (32 * m < hr && hist_row);       // rows lj + 32 m below hr: history
		if (from_ring) d[m] = ringl[(p.win_base + (long) (lj + 32 * m) * N2 + (col0 + (long) it * TW) + lt) & p.ring_mask];
		else if (32 * m < hr) d[m] = slab_ld(vs0, so);                                 // (hr < 32: the rows hr .. 31 of m = 0)
		else d[m] = slab_ld(vs, so + (int) __builtin_amdgcn_readfirstlane((int) ((32 * m - hr) * N2 * fb)));
	};
	auto fetch = [&](int it, cplx (&d)[PT]) {
		const int so = tile_so(it);
#pragma unroll
		for (int m = 0; m < PT; ++m) fetch1(it, so, m, d);
	};
	auto stage = [&](const cplx (&d)[PT], cplx *dst) {   // a fetched tile into a tile buffer
#pragma unroll
		for (int m = 0; m < PT; ++m) dst[lq * FZ_QS + (lj + P * m) * FZ_PITCH + lt] = d[m];
	};
	// one section table per pair (f.sec_stride != 0): a wave is 64 rows of ONE pair, so its table is still read with scalar loads
	sec += (size_t) __builtin_amdgcn_readfirstlane((int) ((2 * grp + rq) * f.sec_stride));
	const double gn = f.gain_tab ? f.gain_tab[__builtin_amdgcn_readfirstlane(2 * grp + rq)] : f.gain;
	// states of this thread's row: chunk (rr - HR) seg + sg of channels 4 grp + 2 rq, + 1
	double2 m0[NSEC], m1[NSEC];
	const bool rec = rr >= hr;                       // (history rows pass through unchanged)
	{
		const int D = 2 * f.n_ops;
		const long c = rec ? (long) (rr - hr) * f.seg + sg : 0;
		const double *xs = f.X + (((size_t) s * f.K + c) * f.C + 4 * grp + 2 * rq) * D;
#pragma unroll
		for (int k = 0; k < NSEC; ++k) {
			const int op = f.sec_op[k];
			if (op >= 0 && rec) {
				m0[k] = make_double2(xs[2 * op], xs[D + 2 * op]);
				m1[k] = make_double2(xs[2 * op + 1], xs[D + 2 * op + 1]);
			}
			else { m0[k] = make_double2(0.0, 0.0); m1[k] = make_double2(0.0, 0.0); }
		}
	}
	const bool keeps = rr >= N1 - hr;                // this row is history of the next window
	double2 *ringw = const_cast<double2 *>(ring0) + rq * p.ring_row_stride;
	const long ring_e0 = p.win_base + (long) rr * N2 + col0;      // (win_base is a multiple of 8 here: a run of 8 never straddles the ring's end)
	cplx nx[PT];
	fetch(0, nx);
	stage(nx, buf0);
	SecCoef cf = load_coef(sec, 0);
	lds_barrier();                                   // twiddle table and tile 0 visible
	const FzTw tw{ twt };
	const long tw_row = (long) (j & 15) * p.N2;
	// Per tile (three barriers): the thread's row through the sections, in place in `cur` | the column's points out of `cur`, the next
	// tile's frames into the other buffer, pass 1 in place | pass 2 in place | pass 3, twiddle, stores.  What orders the rest: a wave
	// stages into the other buffer only behind this tile's first barrier, which every wave reaches after its last look at that buffer
	// (the previous tile's third pass); the staged frames are read behind this tile's second barrier at the earliest.
	for (int it = 0; it < tiles; ++it) {
		cplx *cur = buf0 + (it & 1) * 2 * FZ_QS, *nxt = buf0 + ((it + 1) & 1) * 2 * FZ_QS;
		const FzMap1 map1{ q * FZ_QS + t };
		const FzMap2 map2{ q * FZ_QS + t };
		// the next tile's frames, asked for one row set per section of the recurrence (the last iteration re-reads its own tile: the loop
		// body stays uniform), and this tile's inter-pass twiddles (two table entries)
		const int nit = it + 1 < tiles ? it + 1 : it;
		const int nso = tile_so(nit);
		if constexpr (NSEC < PT) fetch(nit, nx);
		const long col = col0 + (long) it * TW;
		const cplx tw_s = TAB(p.tw_col)[(long) 16 * p.N2 + col + t], tw_a = TAB(p.tw_col)[tw_row + col + t];
		// this thread's row: 8 consecutive frames of its pair through the sections, back into the same slots (history rows stay as staged)
		{
			cplx x[TW];
#pragma unroll
			for (int i = 0; i < TW; ++i) x[i] = cur[rq * FZ_QS + rr * FZ_PITCH + i];
			run_sections<NSEC, TW>(x, m0, m1, sec, cf, [&](int k) { if constexpr (NSEC >= PT) { if (k < PT) fetch1(nit, nso, k, nx); } });
			if (gn != 1.0) {
#pragma unroll
				for (int i = 0; i < TW; ++i) { x[i].x *= gn; x[i].y *= gn; }
			}
			if (rec) {                                   // (divergent in the first wave of either pair only)
#pragma unroll
				for (int i = 0; i < TW; ++i) cur[rq * FZ_QS + rr * FZ_PITCH + i] = x[i];
			}
			if (keeps) {                                 // the rows the next window looks back at also go to the rings
				double2 *w0 = ringw + ((ring_e0 + (long) it * TW) & p.ring_mask);
#pragma unroll
				for (int i = 0; i < TW; ++i) w0[i] = x[i];
			}
		}
		lds_barrier();                                   // the cascade's output visible
		cplx v[PT];
#pragma unroll
		for (int m = 0; m < PT; ++m) v[m] = cur[q * FZ_QS + (j + P * m) * FZ_PITCH + t];
		stage(nx, nxt);                                  // the next tile's frames into the other buffer
		fz_pass<PT, 8, 8, 1, false>(v, j, cur, map1, tw);    // (in place: a thread's outputs go where its inputs came from)
		lds_barrier();
		gather_n<PT>(v, j, cur, map1);
		fz_pass<PT, 8, 8, 8, false>(v, j, cur, map2, tw);
		lds_barrier();
		gather_n<PT>(v, j, cur, map2);
		fz_pass<PT, 8, 4, 64, true>(v, j, cur, map2, tw);
		fz_twiddle(tw_s, tw_a, j, v);
		const int wo = vw + (int) (col * (long) sizeof(cplx));
#pragma unroll
		for (int m = 0; m < PT; ++m) buf_stc<2>(v[m], rw, wo + m * w_step);
	}
}

template <int NSEC> static void launch_pre(const FuseParams &f, const double *sec, long N2, int pps, hipStream_t st)
{
	const long n = f.K * pps;
	const dim3 grid((unsigned) ((n + 255) / 256), (unsigned) f.n_streams);
	if (f.sec_stride) hipLaunchKernelGGL((fused_prepass<NSEC, true>), grid, dim3(256), 0, st, f, sec, N2, pps);
	else hipLaunchKernelGGL((fused_prepass<NSEC, false>), grid, dim3(256), 0, st, f, sec, N2, pps);
}

template <int NSEC, int HR> static void launch_col(const ConvParams &p, const FuseParams &f, const double *sec, hipStream_t st)
{
	const unsigned wgs = (unsigned) ((long) f.n_streams * (p.pairs_per_stream / 2) * f.seg);
	if (f.in_fmt != PCM_DOUBLE) {
		if (f.in_fmt == PCM_S16) {
			grant_dynamic_lds(reinterpret_cast<const void *>(fused_col_fwd<NSEC, HR, 2>), FZ_LDS);
			hipLaunchKernelGGL((fused_col_fwd<NSEC, HR, 2>), dim3(wgs), dim3(512), FZ_LDS, st, p, f, sec);
		}
		else {
			grant_dynamic_lds(reinterpret_cast<const void *>(fused_col_fwd<NSEC, HR, 4>), FZ_LDS);
			hipLaunchKernelGGL((fused_col_fwd<NSEC, HR, 4>), dim3(wgs), dim3(512), FZ_LDS, st, p, f, sec);
		}
		return;
	}
	grant_dynamic_lds(reinterpret_cast<const void *>(fused_col_fwd<NSEC, HR>), FZ_LDS);
	hipLaunchKernelGGL((fused_col_fwd<NSEC, HR>), dim3(wgs), dim3(512), FZ_LDS, st, p, f, sec);
}

template <int NSEC> static bool launch_col_mh(const ConvParams &p, const FuseParams &f, const double *sec, hipStream_t st)
{
	switch (f.hist_rows) {
	case 16: launch_col<NSEC, 16>(p, f, sec, st); return true;
	case 17: launch_col<NSEC, 17>(p, f, sec, st); return true;   // (BASELINE config 4: 66119 taps of fir_p merged into the 2x resampler; 8.54 against 8.92 ms for the run-time instance, profiles/r05_config4_hr17.txt)
	case 32: launch_col<NSEC, 32>(p, f, sec, st); return true;
	default:
		if (f.hist_rows < 1 || f.hist_rows > 32) return false;
		launch_col<NSEC, 0>(p, f, sec, st);
		return true;
	}
}

}  // namespace pfz

// the matrix-core form of the prepass: 8 channels per stream, chunks in groups of 8, at most 16 sections' states; Gt: [len][32] (fuse_gtable)
bool launch_fused_prepass_mm(const FuseParams &f, const double *Gt, long N2, int n_state, hipStream_t st)
{
	if (f.C != 8 || f.K < 1 || (f.len % 8) != 0 || n_state < 1 || n_state > 32 || f.sec_stride != 0) return false;      // (one G for every column of the product)
	const dim3 grid((unsigned) (((f.K + 7) / 8 + 3) / 4), (unsigned) f.n_streams);
	if (f.in_fmt != PCM_DOUBLE) {
		if (!pcm_fusable(f.in_fmt)) return false;
		if (f.in_fmt == PCM_S16) {
			if (n_state <= 16) hipLaunchKernelGGL((pfz::fused_prepass_mm<1, 2>), grid, dim3(256), 0, st, f, Gt, N2, n_state);
			else hipLaunchKernelGGL((pfz::fused_prepass_mm<2, 2>), grid, dim3(256), 0, st, f, Gt, N2, n_state);
		}
		else if (n_state <= 16) hipLaunchKernelGGL((pfz::fused_prepass_mm<1, 4>), grid, dim3(256), 0, st, f, Gt, N2, n_state);
		else hipLaunchKernelGGL((pfz::fused_prepass_mm<2, 4>), grid, dim3(256), 0, st, f, Gt, N2, n_state);
		return true;
	}
	if (n_state <= 16) hipLaunchKernelGGL((pfz::fused_prepass_mm<1>), grid, dim3(256), 0, st, f, Gt, N2, n_state);
	else hipLaunchKernelGGL((pfz::fused_prepass_mm<2>), grid, dim3(256), 0, st, f, Gt, N2, n_state);
	return true;
}

// section counts with an instance (a chain with another count is padded with pass-through sections by the host: fuse_sections)
int fused_section_slots(int n_sec)
{
	for (int n : { 1, 2, 4, 6, 8, 10, 12 }) if (n_sec <= n) return n;
	return 0;
}

bool launch_fused_prepass(const FuseParams &f, const double *sec, long N2, int pps, hipStream_t st)
{
	switch (f.n_sec) {
	case 1: pfz::launch_pre<1>(f, sec, N2, pps, st); return true;
	case 2: pfz::launch_pre<2>(f, sec, N2, pps, st); return true;
	case 4: pfz::launch_pre<4>(f, sec, N2, pps, st); return true;
	case 6: pfz::launch_pre<6>(f, sec, N2, pps, st); return true;
	case 8: pfz::launch_pre<8>(f, sec, N2, pps, st); return true;
	case 10: pfz::launch_pre<10>(f, sec, N2, pps, st); return true;
	case 12: pfz::launch_pre<12>(f, sec, N2, pps, st); return true;
	default: return false;
	}
}

bool launch_fused_col_fwd(const ConvParams &p, const FuseParams &f, const double *sec, hipStream_t st)
{
	if (p.log2N1 != 8 || (p.pairs_per_stream & 1)) return false;
	switch (f.n_sec) {
	case 1: return pfz::launch_col_mh<1>(p, f, sec, st);
	case 2: return pfz::launch_col_mh<2>(p, f, sec, st);
	case 4: return pfz::launch_col_mh<4>(p, f, sec, st);
	case 6: return pfz::launch_col_mh<6>(p, f, sec, st);
	case 8: return pfz::launch_col_mh<8>(p, f, sec, st);
	case 10: return pfz::launch_col_mh<10>(p, f, sec, st);
	case 12: return pfz::launch_col_mh<12>(p, f, sec, st);
	default: return false;
	}
}

}  // namespace dspamd
