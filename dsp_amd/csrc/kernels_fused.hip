// kernels_fused.hip -- the biquad cascade fused into the FFT convolver's first pass (round 4).
//
// What it replaces, per block, in the reference: the passes of biquad_effect_run (biquad.c:296-315, one per section, with
// biquad() of biquad.h:76-92 inside) followed by the input half of fir_p's block transform (fir_p.c:64-103) -- here ONE
// trip of the samples through HBM instead of two (cascade_rows writing the pair rings + conv_col_fwd reading them back).
//
// The four-step transform's first pass is the STRIDED one (the column of a window holds samples N2 frames apart), and a
// recurrence wants consecutive frames.  Both are served by giving a thread a whole ROW of the window: thread r of a workgroup
// owns the N2 consecutive frames of row r and walks along them 16 at a time, its section states in registers; the 256 rows of
// a tile of 16 columns are then exactly 16 complete columns, transformed on the spot (the two radix-16 passes of conv_col_fwd).
// Rows are 4096 frames apart in time, so the state a row starts from is not known to the kernel that walks all rows at once:
// a first kernel (fused_prepass) runs every row from ZERO state and keeps only the END state, cascade_chunk_carry
// (kernels_chunk.hip: x' = M x + e over the rows, M = A^N2 from the host in extended precision) turns those into the true state at
// every row start, and fused_col_fwd runs the recurrence again from those states -- the reference's own recurrence, sample by
// sample, no correction terms.  The recurrence costs its 5 fp64 operations per sample and section twice (cascade_rows spends
// 12 once: zero-state pass, row scan, zero-input correction) and the input is read twice; the cascade's output never goes to
// HBM except for the last `first_n` frames of the call, which are the next window's history (the pair rings, as before).
//
// Layout of fused_col_fwd: workgroup = (stream, group of 2 channel pairs, row segment), 256 threads = one wave per SIMD with the
// whole register file (state 4 channels x NSEC x 2, two register sets of 2 x 16 complex points: about 420 of the 512);
// LDS = [2 pairs][256 rows][16 columns] complex, pitch 17 (a thread's own row: conflict-free both ways), reused as the
// exchange buffer of the column transform.  The two groups of a stream are dispatched 8 workgroup ids apart (same XCD, same L2):
// each owns 32 of the 64 bytes of every frame.
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
template <int NSEC, int NF>
__device__ __forceinline__ void run_sections(cplx (&x)[NF], double2 (&m0)[NSEC], double2 (&m1)[NSEC], const double *__restrict__ sec)
{
#pragma unroll
	for (int k = 0; k < NSEC; ++k) {
		const double c0 = sec[6 * k], c1 = sec[6 * k + 1], c2 = sec[6 * k + 2], c3 = sec[6 * k + 3], c4 = sec[6 * k + 4];
		double a0x = m0[k].x, a0y = m0[k].y, a1x = m1[k].x, a1y = m1[k].y;
#pragma unroll
		for (int i = 0; i < NF; ++i) {
			const double sa = x[i].x, sb = x[i].y;
			const double ra = fma(c0, sa, a0x), rb = fma(c0, sb, a0y);
			a0x = fma(-c3, ra, fma(c1, sa, a1x)); a0y = fma(-c3, rb, fma(c1, sb, a1y));
			a1x = fma(-c4, ra, c2 * sa); a1y = fma(-c4, rb, c2 * sb);
			x[i].x = ra; x[i].y = rb;
		}
		m0[k].x = a0x; m0[k].y = a0y; m1[k].x = a1x; m1[k].y = a1y;
	}
}

// ---- pass 0: the end state of every chunk run from zero state.  grid (ceil(K pps / 256), S): thread = (chunk c, pair q) of
// stream blockIdx.y, pair-fastest so that the lanes of one chunk read whole frames; every thread walks its len frames, 16 at a
// time, the next 16 already on their way.
template <int NSEC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fused_prepass(FuseParams f, const double *__restrict__ sec, long N2, int pps)
{
	const long id = (long) blockIdx.x * 256 + threadIdx.x;
	const long s = blockIdx.y;
	const bool live = id < f.K * pps;
	const int q = (int) (id % pps);
	const long c = live ? id / pps : 0;
	const long row = c / f.seg;
	const int sg = (int) (c - row * f.seg);
	const long frame0 = row * N2 + (long) sg * f.len;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(f.in) + (size_t) s * f.in_stride_frames * f.C, 0, 0x7fffffff, 0x00020000);
	const int fb = f.C * (int) sizeof(double);                      // bytes per frame
	int vo = (int) ((frame0 * f.C + 2 * q) * (long) sizeof(double));
	double2 m0[NSEC], m1[NSEC];
#pragma unroll
	for (int k = 0; k < NSEC; ++k) { m0[k] = make_double2(0.0, 0.0); m1[k] = make_double2(0.0, 0.0); }
	cplx x[16], nx[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) nx[i] = buf_ldc(rs, vo, i * fb);
	for (long i0 = 0; i0 < f.len; i0 += 16) {
#pragma unroll
		for (int i = 0; i < 16; ++i) x[i] = nx[i];
		if (i0 + 16 < f.len) {
			vo += 16 * fb;
#pragma unroll
			for (int i = 0; i < 16; ++i) nx[i] = buf_ldc(rs, vo, i * fb);
		}
		run_sections<NSEC, 16>(x, m0, m1, sec);
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

// ---- pass 1: K1 with the cascade in front of its column transforms (see the head of the file).
// grid: S x groups x seg workgroups in the order of fz_block (the groups of a stream on one XCD), 256 threads = 4 waves, one per SIMD.
// MH = hist_rows / 16: the window rows j + 16 m with m < MH are history (the pair rings), the others new frames (the slab).
// A tile = 8 columns of the group's two pairs.  Two LDS buffers [2 pairs][256 rows] of pitch 9: `raw` holds the frames of the tile
// as loaded (written in the column layout (pair, column, j), read by rows), `yb` the cascade's output (written by rows, read in
// the column layout) and then the exchange of the two radix-16 passes; the next tile's frames are loaded into registers under
// the recurrence of this one and move to `raw` before its transform starts.
constexpr int FZ_TW = 8, FZ_PITCH = 9, FZ_QS = 256 * FZ_PITCH + 4;     // (the two pairs' rows 4 slots apart mod 8: the 8 lanes of a write group are (pair, column 0..3))
constexpr int FZ_XQS = 256 * FZ_TW + 8;                                 // the exchange layout: ColCfg<8, 2>::QS
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

template <int NSEC, int MH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void fused_col_fwd(ConvParams p, FuseParams f, const double *__restrict__ sec)
{
	constexpr int N1 = 256, P = 16, HR = 16 * MH, TW = FZ_TW;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	cplx *raw = reinterpret_cast<cplx *>(smem_raw);
	cplx *yb = raw + 2 * FZ_QS;
	cplx *twt = yb + 2 * FZ_QS;
	const int tid = threadIdx.x;
	const int q = tid & 1, t = (tid >> 1) & (TW - 1), j = tid >> 4;      // column transform: points n1 = j + 16 m of column t of pair q
	const int rr = tid;                                                   // recurrence: row rr of the window, both pairs
	const int groups = p.pairs_per_stream >> 1;
	int s, gs;
	fz_block(f.n_streams, groups * f.seg, s, gs);
	const int grp = gs % groups, sg = gs / groups;
	twt[tid] = TAB(p.tw_n1)[tid];
	const long N2 = p.N2;
	const int tiles = (int) (N2 / TW / f.seg);
	const long col0 = (long) sg * tiles * TW;        // first column of this workgroup's segment
	const long pair0 = (long) s * p.pairs_per_stream + 2 * grp;     // the group's first pair
	// the slab through a buffer descriptor: element (row j + 16 m >= HR, column t) of pair q at vs + ((m - MH) 16 N2 + column) frame bytes
	const int fb = f.C * (int) sizeof(double);
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(f.in) + (size_t) s * f.in_stride_frames * f.C, 0, 0x7fffffff, 0x00020000);
	const int vs = (int) ((((long) j * N2 + t) * f.C + 4 * grp + 2 * q) * (long) sizeof(double));
	const int row_step = (int) (16 * N2 * fb);       // 16 rows further
	// W of the group's two pairs through one descriptor
	const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(WBUF(p.W) + (pair0 - p.pair0) * p.w_stride, 0, 0x7fffffff, 0x00020000);
	const int vw = (int) (((long) q * p.w_stride + (long) j * N2 + t) * (long) sizeof(cplx));
	const int w_step = (int) (16 * N2 * (long) sizeof(cplx));
	const double2 *ring0 = p.ring + pair0 * p.ring_row_stride;
	const double2 *ringq = ring0 + q * p.ring_row_stride;
	auto fetch = [&](int it, cplx (&d)[16]) {
		const long col = col0 + (long) it * TW;
		const int so = (int) (col * fb);
#pragma unroll
		for (int m = 0; m < 16; ++m) {
			if (m < MH) d[m] = ringq[(p.win_base + (long) (j + 16 * m) * N2 + col + t) & p.ring_mask];
			else d[m] = buf_ldc(rs, vs, so + (m - MH) * row_step);
		}
	};
	// states of this thread's row: chunk (rr - HR) seg + sg of channels 4 grp .. 4 grp + 3
	double2 m0[2][NSEC], m1[2][NSEC];
	const bool rec = rr >= HR;
	{
		const int D = 2 * f.n_ops;
		const long c = rec ? (long) (rr - HR) * f.seg + sg : 0;
		const double *xs = f.X + (((size_t) s * f.K + c) * f.C + 4 * grp) * D;
#pragma unroll
		for (int qq = 0; qq < 2; ++qq)
#pragma unroll
			for (int k = 0; k < NSEC; ++k) {
				const int op = f.sec_op[k];
				if (op >= 0 && rec) {
					m0[qq][k] = make_double2(xs[(2 * qq) * D + 2 * op], xs[(2 * qq + 1) * D + 2 * op]);
					m1[qq][k] = make_double2(xs[(2 * qq) * D + 2 * op + 1], xs[(2 * qq + 1) * D + 2 * op + 1]);
				}
				else { m0[qq][k] = make_double2(0.0, 0.0); m1[qq][k] = make_double2(0.0, 0.0); }
			}
	}
	const bool keeps = rr >= N1 - HR;                // this row is history of the next window
	double2 *ringw = const_cast<double2 *>(ring0);
	const long ring_e0 = p.win_base + (long) rr * N2 + col0;      // (win_base is a multiple of 8 here: a run of 8 never straddles the ring's end)
	cplx nx[16], hist[MH], hist_next[MH];
	fetch(0, nx);
#pragma unroll
	for (int m = 0; m < 16; ++m) { if (m < MH) hist[m] = nx[m]; else raw[q * FZ_QS + (j + 16 * m) * FZ_PITCH + t] = nx[m]; }
	lds_barrier();                                   // twiddle table and tile 0 visible
	const TwCol tw{ twt };
	const ColMap<TW> xmap{ q * FZ_XQS + t };
	for (int it = 0; it < tiles; ++it) {
		fetch(it + 1 < tiles ? it + 1 : it, nx);         // (the last iteration re-reads its own tile: the loop body stays uniform)
		// this thread's row: 8 consecutive frames of both pairs through the sections
		{
			cplx x0[TW], x1[TW];
#pragma unroll
			for (int i = 0; i < TW; ++i) {
				x0[i] = rec ? raw[rr * FZ_PITCH + i] : mkc(0.0, 0.0);
				x1[i] = rec ? raw[FZ_QS + rr * FZ_PITCH + i] : mkc(0.0, 0.0);
			}
			run_sections<NSEC, TW>(x0, m0[0], m1[0], sec);
			run_sections<NSEC, TW>(x1, m0[1], m1[1], sec);
			if (f.gain != 1.0) {
#pragma unroll
				for (int i = 0; i < TW; ++i) { x0[i].x *= f.gain; x0[i].y *= f.gain; x1[i].x *= f.gain; x1[i].y *= f.gain; }
			}
			// the cascade's output into the second buffer; the rows the next window looks back at also go to the rings
			if (rec) {
#pragma unroll
				for (int i = 0; i < TW; ++i) { yb[rr * FZ_PITCH + i] = x0[i]; yb[FZ_QS + rr * FZ_PITCH + i] = x1[i]; }
			}
			if (keeps) {
				double2 *w0 = ringw + ((ring_e0 + (long) it * TW) & p.ring_mask);
#pragma unroll
				for (int i = 0; i < TW; ++i) { w0[i] = x0[i]; w0[p.ring_row_stride + i] = x1[i]; }
			}
		}
		lds_barrier();                                   // output visible; every row of `raw` has been read
		// the next tile's frames into `raw`
#pragma unroll
		for (int m = 0; m < 16; ++m) { if (m < MH) hist_next[m] = nx[m]; else raw[q * FZ_QS + (j + 16 * m) * FZ_PITCH + t] = nx[m]; }
		cplx v[16];
#pragma unroll
		for (int m = 0; m < 16; ++m) v[m] = (m < MH) ? hist[m] : yb[q * FZ_QS + (j + 16 * m) * FZ_PITCH + t];
#pragma unroll
		for (int m = 0; m < MH; ++m) hist[m] = hist_next[m];
		lds_barrier();                                   // every thread has its points: `yb` becomes the exchange buffer
		pass16<8, 16, 1, false, false>(v, j, yb, xmap, tw);
		lds_barrier();
		gather16<8>(v, j, yb, xmap);
		pass16<8, 16, 16, false, true>(v, j, yb, xmap, tw);
		const long col = col0 + (long) it * TW;
		col_twiddle<false, P>(p, col + t, j, v);
		const int wo = vw + (int) (col * (long) sizeof(cplx));
#pragma unroll
		for (int m = 0; m < 16; ++m) buf_stc<2>(v[m], rw, wo + m * w_step);
		lds_barrier();                                   // the exchange reads are done before the next output is written; `raw` visible
	}
}

template <int NSEC> static void launch_pre(const FuseParams &f, const double *sec, long N2, int pps, hipStream_t st)
{
	const long n = f.K * pps;
	hipLaunchKernelGGL((fused_prepass<NSEC>), dim3((unsigned) ((n + 255) / 256), (unsigned) f.n_streams), dim3(256), 0, st, f, sec, N2, pps);
}

template <int NSEC, int MH> static void launch_col(const ConvParams &p, const FuseParams &f, const double *sec, hipStream_t st)
{
	grant_dynamic_lds(reinterpret_cast<const void *>(fused_col_fwd<NSEC, MH>), FZ_LDS);
	const unsigned wgs = (unsigned) ((long) f.n_streams * (p.pairs_per_stream / 2) * f.seg);
	hipLaunchKernelGGL((fused_col_fwd<NSEC, MH>), dim3(wgs), dim3(256), FZ_LDS, st, p, f, sec);
}

template <int NSEC> static bool launch_col_mh(const ConvParams &p, const FuseParams &f, const double *sec, hipStream_t st)
{
	switch (f.hist_rows) {
	case 16: launch_col<NSEC, 1>(p, f, sec, st); return true;
	case 32: launch_col<NSEC, 2>(p, f, sec, st); return true;
	default: return false;
	}
}

}  // namespace pfz

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
