// kernels_cascade.hip -- fused gain/add/biquad cascade as a block-parallel state-space recurrence.
//
// Replaces, for a whole batch of streams in one launch, the reference's per-effect passes
//   gain_effect_run / add_effect_run        gain.c:25-43
//   biquad_effect_run{,_all} + biquad()     biquad.c:296-315, biquad.h:76-92
// (the reference never merges biquads that act on the same channel, biquad.c:344-351, so a chain of
// 10 biquads is 10 full passes over the block there; here it is one read and one write).
//
// Four kernels share the arithmetic below and differ in how they spread it over the chip (launch_cascade picks):
//   cascade_rows   a wave = 4 channels (one per DPP row), 32 frames per lane, time axis shared by skewed waves -- many
//                  channels with identical sections (the headline kernel)
//   (cascade_wave, one channel per wave with the time axis shared by up to 10 skewed waves, served few channels until the chunked time axis did: removed in round 6)
//   cascade_fast   a wave = 1 channel, cooperative workgroup I/O -- per-channel coefficients, add, unselected channels
//   cascade_kernel generic: any channel count, remainders shorter than a tile (L = 4 and L = 1 tail steps)
//
// Math.  One TDF-II section with x = (m0, m1):
//     r[n] = c0 s[n] + m0[n],   x[n+1] = A x[n] + B s[n],   A = [[-c3, 1], [-c4, 0]]
// A wave owns 64*L consecutive samples of one channel, lane l owning samples [lL, lL+L):
//   1. every lane runs the recurrence from ZERO state over its L samples (registers only);
//   2. the lane end-states b_l are combined with a 6-step Kogge-Stone scan over the wave using the
//      constant matrices A^(L 2^k) (the carried state enters through lane 0: b_0 += A^L x_in);
//   3. every lane adds the zero-input response of its true incoming state x_l, r[i] += (A^i x_l)[0],
//      obtained by running the homogeneous recurrence (2 FMA per sample, no tables).
// Sections run back to back on the register-resident samples; the state leaving lane 63 is the state
// carried to the next tile / next run() call, exactly the reference's (m0, m1).
// Blocks whose length is not a multiple of 64*L finish with L = 1 steps (64 samples per wave step,
// cut at any sample), so consecutive calls of ANY size form one stream.
//
// Layout: workgroup = (stream, channel group); the [frames][C] slab of the stream is read coalesced
// (16 B per lane when the channel count allows), transposed through LDS (row stride padded so that both
// the transposing ds_write and the per-lane ds_read_b64 are bank-conflict free), results go back the same
// way and/or into the pair ring that feeds the FFT convolver.  The next tile's global loads are issued
// before the current tile's recurrences so HBM latency hides under the fp64 work; the per-(channel, op)
// coefficients are staged in LDS once per launch and read as broadcasts.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "kparams.h"
#include "pcm_device.h"

namespace dspamd {

constexpr int L16 = CASCADE_L;
// per-channel LDS row: 64 lanes x (16 + 1 pad) doubles, + 2 so that rows of different channels land on
// different bank pairs for the transposing store (see docs/history.md section 4.1)
constexpr int CH_STRIDE = 64 * (L16 + 1) + 2;
constexpr int NPOW_USED = 10;              // A^(2^k), k = 0..9
constexpr int OPL_DOUBLES = 8 + 4 * NPOW_USED;   // compact LDS descriptor: kind, g, c0..c4, pad, P[10][4]
constexpr int MAX_PF = 16;                 // prefetch registers (doubles) per thread

__device__ __forceinline__ int lds_index(int t) { return t + (t >> 4); }

// Which (stream, channel group) a workgroup of a (streams, groups) grid works on.  The channel groups of a stream share every
// 64-byte frame of its slabs (a group of 4 channels owns 32 bytes of it), so they should run at the same time on the same XCD:
// its L2 then sees whole lines -- fetched once, written back whole.  Run a whole dispatch wave apart (the plain x-fastest
// order), each group fetches the lines again and its half lines go to HBM as partial writes: a chain that ENDS in sections
// (interleaved output) 18.9 -> 8.5 ms at 256 x 8 channels, the headline's cascade (ring output) 9.5 -> 8.6 ms.  Workgroups are
// dispatched in linear id order (x fastest) and dealt to the 8 XCDs round robin: ids that differ by 8 meet in one L2, a few
// dispatch slots apart -- blocks of 8 streams, group-major inside.
__device__ __forceinline__ void stream_and_group_of_block(int xcd_map, int &s, int &grp)
{
	const int ns = gridDim.x, ng = gridDim.y;
	if (xcd_map && ng > 1 && (ns & 7) == 0) {
		const int id = blockIdx.x + blockIdx.y * ns, per = 8 * ng, blk = id / per, r = id % per;
		s = blk * 8 + (r & 7);
		grp = r >> 3;
	}
	else { s = blockIdx.x; grp = blockIdx.y; }
}

// ---- cross-lane helpers: DPP moves of fp64 values (two 32-bit DPP ops), no LDS traffic ----
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
	// bound_ctrl = true: lanes whose source is outside the row / wave read 0.0
	const long long b = __double_as_longlong(v);
	const int lo = __builtin_amdgcn_update_dpp(0, (int) (b & 0xffffffffLL), CTRL, 0xf, 0xf, true);
	const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), CTRL, 0xf, 0xf, true);
	return __longlong_as_double(((long long) hi << 32) | (unsigned int) lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
	const long long b = __double_as_longlong(v);
	const int lo = __builtin_amdgcn_readlane((int) (b & 0xffffffffLL), lane);
	const int hi = __builtin_amdgcn_readlane((int) (b >> 32), lane);
	return __longlong_as_double(((long long) hi << 32) | (unsigned int) lo);
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118, DPP_WAVE_SHR1 = 0x138;

// inclusive scan WITHIN each 16-lane row of the affine recurrence x <- P x + b (P constant, given as powers
// P^(2^k) at Pw[4k..4k+3], k = 0..3): m_l = sum_{j <= l, same row} P^(l-j) b_j
__device__ __forceinline__ void row_scan(double &m0, double &m1, const double (&Pw)[16])
{
	double t0, t1;
	t0 = dpp_f64<DPP_ROW_SHR1>(m0); t1 = dpp_f64<DPP_ROW_SHR1>(m1);
	m0 += Pw[0] * t0 + Pw[1] * t1; m1 += Pw[2] * t0 + Pw[3] * t1;
	t0 = dpp_f64<DPP_ROW_SHR2>(m0); t1 = dpp_f64<DPP_ROW_SHR2>(m1);
	m0 += Pw[4] * t0 + Pw[5] * t1; m1 += Pw[6] * t0 + Pw[7] * t1;
	t0 = dpp_f64<DPP_ROW_SHR4>(m0); t1 = dpp_f64<DPP_ROW_SHR4>(m1);
	m0 += Pw[8] * t0 + Pw[9] * t1; m1 += Pw[10] * t0 + Pw[11] * t1;
	t0 = dpp_f64<DPP_ROW_SHR8>(m0); t1 = dpp_f64<DPP_ROW_SHR8>(m1);
	m0 += Pw[12] * t0 + Pw[13] * t1; m1 += Pw[14] * t0 + Pw[15] * t1;
}

// ops: LDS descriptors of this channel: [n_ops][OPD] doubles = kind, g, c0..c4, pad, then matrices; the five
// matrices P^(L), P^(2L), P^(4L), P^(8L), P^(16L) start at od[PW_OFF]
template <int L, int OPD, int PW_OFF>
__device__ __forceinline__ void run_ops(double (&v)[L], const double *ops, int n_ops, double *st /* LDS [n_ops][2] */,
                                        int lane, int last_lane)
{
	const int row = lane >> 4;
	for (int j = 0; j < n_ops; ++j) {
		const double *od = ops + j * OPD;
		const int kind = __double_as_longlong(od[0]);
		if (kind == OP_MUL) {
			const double g = od[1];
#pragma unroll
			for (int i = 0; i < L; ++i) v[i] = __dmul_rn(v[i], g);
		}
		else if (kind == OP_ADD) {
			const double g = od[1];
#pragma unroll
			for (int i = 0; i < L; ++i) v[i] = __dadd_rn(v[i], g);
		}
		else if (kind == OP_BIQUAD) {
			// all constants of the section up front: their LDS latency hides under the recurrence below
			const double c0 = od[2], c1 = od[3], c2 = od[4], nc3 = -od[5], nc4 = -od[6];
			const double xin0 = st[2*j], xin1 = st[2*j + 1];
			double Pw[16], P16[4];
#pragma unroll
			for (int i = 0; i < 16; ++i) Pw[i] = od[PW_OFF + i];          // P^(L), P^(2L), P^(4L), P^(8L)
#pragma unroll
			for (int i = 0; i < 4; ++i) P16[i] = od[PW_OFF + 16 + i];    // P^(16L): one whole row
			double m0 = 0.0, m1 = 0.0;
#pragma unroll
			for (int i = 0; i < L; ++i) {
				const double s = v[i];
				const double r = fma(c0, s, m0);
				m0 = fma(nc3, r, fma(c1, s, m1));
				m1 = fma(nc4, r, c2 * s);
				v[i] = r;
			}
			if (lane == 0) {  // the carried state rides through lane 0's L samples
				m0 += Pw[0] * xin0 + Pw[1] * xin1;
				m1 += Pw[2] * xin0 + Pw[3] * xin1;
			}
			// pass 1: per-row inclusive scans; row totals sit in lanes 15, 31, 47, 63
			row_scan(m0, m1, Pw);
			const double T00 = readlane_f64(m0, 15), T01 = readlane_f64(m1, 15);
			const double T10 = readlane_f64(m0, 31), T11 = readlane_f64(m1, 31);
			const double T20 = readlane_f64(m0, 47), T21 = readlane_f64(m1, 47);
			// true states at the row boundaries: E_r = P^(16L) E_(r-1) + T_r
			const double E10 = P16[0] * T00 + P16[1] * T01 + T10, E11 = P16[2] * T00 + P16[3] * T01 + T11;
			const double E20 = P16[0] * E10 + P16[1] * E11 + T20, E21 = P16[2] * E10 + P16[3] * E11 + T21;
			// pass 2: carry c_r = E_(r-1) enters each row at its first lane as P^L c_r and is spread by the same scan
			const double cr0 = (row == 1) ? T00 : (row == 2) ? E10 : E20;
			const double cr1 = (row == 1) ? T01 : (row == 2) ? E11 : E21;
			double u0 = 0.0, u1 = 0.0;
			if (row > 0 && (lane & 15) == 0) { u0 = Pw[0] * cr0 + Pw[1] * cr1; u1 = Pw[2] * cr0 + Pw[3] * cr1; }
			row_scan(u0, u1, Pw);
			m0 += u0; m1 += u1;                                  // (m0, m1) = true state after this lane's samples
			double x0 = dpp_f64<DPP_WAVE_SHR1>(m0), x1 = dpp_f64<DPP_WAVE_SHR1>(m1);
			if (lane == 0) { x0 = xin0; x1 = xin1; }
			// zero-input response of the true incoming state
#pragma unroll
			for (int i = 0; i < L; ++i) {
				const double r = x0;
				v[i] += r;
				x0 = fma(nc3, r, x1);
				x1 = nc4 * r;
			}
			// state after the last valid lane's samples = the reference's (m0, m1) at that point
			const double e0 = readlane_f64(m0, last_lane), e1 = readlane_f64(m1, last_lane);
			if (lane == 0) { st[2*j] = e0; st[2*j + 1] = e1; }
		}
	}
}

__global__ __launch_bounds__(512) void cascade_kernel(CascadeParams p, const OpDesc *__restrict__ gops)
{
	extern __shared__ __attribute__((aligned(16))) double smem[];
	int s, grp;
	stream_and_group_of_block(p.xcd_map, s, grp);
	const int c0 = p.cg0 + grp * p.Cg;
	const int cgn = min(p.Cg, p.C - c0);
	const int tid = threadIdx.x, nth = blockDim.x;
	const int lane = tid & 63, wave = tid >> 6, nw = nth >> 6;
	double *tile = smem;                                        // [Cg][CH_STRIDE]
	double *st = tile + (size_t) p.Cg * CH_STRIDE;              // [Cg][n_ops][2]
	double *lops = st + (size_t) p.Cg * p.n_ops * 2;            // [Cg][n_ops][OPL_DOUBLES]

	int cgp = 1, cgs = 0;                                       // next power of two >= cgn
	while (cgp < cgn) { cgp <<= 1; ++cgs; }

	const int n_st = cgn * p.n_ops * 2;
	double *gstate = p.state + ((size_t) s * p.C + c0) * p.n_ops * 2;
	for (int i = tid; i < n_st; i += nth) st[i] = gstate[i];
	for (int i = tid; i < cgn * p.n_ops * OPL_DOUBLES; i += nth) {
		const int q = i % OPL_DOUBLES, co = i / OPL_DOUBLES;     // co = cc * n_ops + j
		const OpDesc *od = gops + (size_t) c0 * p.n_ops + co;
		double v;
		if (q == 0) v = __longlong_as_double((long long) od->kind);
		else if (q == 1) v = od->g;
		else if (q < 7) v = od->c[q - 2];
		else if (q == 7) v = 0.0;
		else v = od->P[(q - 8) >> 2][(q - 8) & 3];
		lops[i] = v;
	}

	// wire formats (in_fmt / sink): the element-wise paths below convert (pcm_device.h); `in` / `out` are then only bases
	const bool wire_in = p.in_fmt != PCM_DOUBLE, sink_on = p.sink.on != 0;
	const long in0 = (long) s * p.in_stride_frames * p.C, out0 = (long) s * p.out_stride_frames * p.C;
	const double *in = p.in + (wire_in ? 0 : in0);
	double *out = p.out + (sink_on ? 0 : out0);
	double peak = 0.0;
	unsigned long long clipped = 0;
	const long n_full = p.frames / CASCADE_TILE;
	const int rem = (int) (p.frames - n_full * CASCADE_TILE);
	const long n_tiles = n_full + (rem ? 1 : 0);
	// full tiles of a group that spans whole frames can be fetched 16 B per lane and prefetched into registers
	const bool vec = (cgn == p.C) && (cgn == cgp) && (cgn >= 2) && ((CASCADE_TILE * cgn) / 2 <= nth * (MAX_PF / 2))
	                 && ((((size_t) in) & 15) == 0) && ((((size_t) out) & 15) == 0) && !wire_in && !sink_on;
	const int npf = vec ? (CASCADE_TILE * cgn / 2 + nth - 1) / nth : 0;   // double2 loads per thread per tile
	double2 pf[MAX_PF / 2];
	if (vec && n_full > 0) {
#pragma unroll
		for (int q = 0; q < MAX_PF / 2; ++q)
			if (q < npf) { const int e = tid + q * nth; if (e < CASCADE_TILE * cgn / 2) pf[q] = reinterpret_cast<const double2 *>(in)[e]; }
	}
	__syncthreads();

	for (long tl = 0; tl < n_tiles; ++tl) {
		const long t0 = tl * CASCADE_TILE;
		const int nfr = (tl < n_full) ? CASCADE_TILE : rem;
		// ---- stage the tile in LDS, transposed ----
		if (vec && nfr == CASCADE_TILE) {
#pragma unroll
			for (int q = 0; q < MAX_PF / 2; ++q) {
				if (q < npf) {
					const int e = tid + q * nth;
					if (e < CASCADE_TILE * cgn / 2) {
						const int t = (2 * e) >> cgs, cc = (2 * e) & (cgp - 1);
						tile[cc * CH_STRIDE + lds_index(t)] = pf[q].x;
						tile[(cc + 1) * CH_STRIDE + lds_index(t)] = pf[q].y;
					}
				}
			}
			if (tl + 1 < n_full) {   // issue the next tile's loads now; they land while this tile computes
				const double2 *nx = reinterpret_cast<const double2 *>(in + (t0 + CASCADE_TILE) * p.C);
#pragma unroll
				for (int q = 0; q < MAX_PF / 2; ++q)
					if (q < npf) { const int e = tid + q * nth; if (e < CASCADE_TILE * cgn / 2) pf[q] = nx[e]; }
			}
		}
		else {
			for (int e = tid; e < (nfr << cgs); e += nth) {
				const int t = e >> cgs, cc = e & (cgp - 1);
				if (cc < cgn)
					tile[cc * CH_STRIDE + lds_index(t)] = wire_in ? pcm_load(p.in, p.in_fmt, in0 + (t0 + t) * p.C + c0 + cc) : in[(t0 + t) * p.C + c0 + cc];
			}
		}
		__syncthreads();
		// ---- recurrences: one wave per channel at a time ----
		for (int cc = wave; cc < cgn; cc += nw) {
			const double *ops = lops + (size_t) cc * p.n_ops * OPL_DOUBLES;
			double *row = tile + cc * CH_STRIDE;
			double *cst = st + cc * p.n_ops * 2;
			if (nfr == CASCADE_TILE) {
				double v[L16];
#pragma unroll
				for (int i = 0; i < L16; ++i) v[i] = row[lane * (L16 + 1) + i];
				run_ops<L16, OPL_DOUBLES, 8 + 4 * 4>(v, ops, p.n_ops, cst, lane, 63);
#pragma unroll
				for (int i = 0; i < L16; ++i) row[lane * (L16 + 1) + i] = v[i];
			}
			else {
				// whole groups of 256 frames as L = 4 steps (a step costs about the same whatever L is: the scans dominate), the
				// rest as L = 1 steps that can be cut at any sample
				int t1 = 0;
				for (; t1 + 256 <= nfr; t1 += 256) {
					double v[4];
#pragma unroll
					for (int i = 0; i < 4; ++i) v[i] = row[lds_index(t1 + 4 * lane + i)];
					run_ops<4, OPL_DOUBLES, 8 + 4 * 2>(v, ops, p.n_ops, cst, lane, 63);
#pragma unroll
					for (int i = 0; i < 4; ++i) row[lds_index(t1 + 4 * lane + i)] = v[i];
				}
				for (; t1 < nfr; t1 += 64) {
					const int t = t1 + lane;
					const int nvalid = min(64, nfr - t1);
					double v[1];
					v[0] = (t < nfr) ? row[lds_index(t)] : 0.0;
					run_ops<1, OPL_DOUBLES, 8>(v, ops, p.n_ops, cst, lane, nvalid - 1);
					if (t < nfr) row[lds_index(t)] = v[0];
				}
			}
		}
		__syncthreads();
		// ---- store ----
		if (p.write_interleaved) {
			if (vec && nfr == CASCADE_TILE) {
				double2 *o2 = reinterpret_cast<double2 *>(out + t0 * p.C);
				for (int e = tid; e < CASCADE_TILE * cgn / 2; e += nth) {
					const int t = (2 * e) >> cgs, cc = (2 * e) & (cgp - 1);
					o2[e] = make_double2(tile[cc * CH_STRIDE + lds_index(t)], tile[(cc + 1) * CH_STRIDE + lds_index(t)]);
				}
			}
			else {
				for (int e = tid; e < (nfr << cgs); e += nth) {
					const int t = e >> cgs, cc = e & (cgp - 1);
					if (cc >= cgn) continue;
					const double v = tile[cc * CH_STRIDE + lds_index(t)];
					if (sink_on) {
						// the pipeline's last kernel: dither, clip and convert on the way out (dsp.c:685-699); remainders only, so the
						// generator values are simply computed per sample
						const long n = (t0 + t) * p.C + c0 + cc;
						const bool dither = p.sink.dither_mult != 0.0;
						uint32_t u0 = 0, u1 = 0;
						if (dither) { u0 = pm_pow<0>((uint64_t) (p.sink.samples_before + n) + 1); u1 = pm_pow<1>((uint64_t) (p.sink.samples_before + n) + 1); }
						pcm_store(p.out, p.sink.fmt, out0 + n, sink_sample(v, dither, u0, u1, p.sink.dither_mult, peak, clipped));
					}
					else out[(t0 + t) * p.C + c0 + cc] = v;
				}
			}
		}
		if (p.ring.base) {
			// the convolver's ring: one 16-byte element (x_a[n], x_b[n]) per frame and channel pair, consecutive lanes =
			// consecutive frames (the workgroup holds every channel of the stream: the feeder requires Cg == C)
			for (int q = 0; q < p.ring.rows_per_stream; ++q) {
				const int ca = p.ring.pair_ch[2 * q] - c0, cb = p.ring.pair_ch[2 * q + 1] - c0;
				double2 *dst = reinterpret_cast<double2 *>(p.ring.base) + ((size_t) s * p.ring.rows_per_stream + q) * p.ring.row_stride;
				for (int t = tid; t < nfr; t += nth) {
					const int li = lds_index(t);
					dst[(p.ring.pos + t0 + t) & p.ring.mask] =
						make_double2((ca >= 0 && ca < cgn) ? tile[ca * CH_STRIDE + li] : 0.0, (cb >= 0 && cb < cgn) ? tile[cb * CH_STRIDE + li] : 0.0);
				}
			}
		}
		__syncthreads();
	}
	if (sink_on && p.sink.stats) sink_stats_block(p.sink.stats, s, peak, clipped);
	for (int i = tid; i < n_st; i += nth) gstate[i] = st[i];
}

// ---------------------------------------------------------------------------------------------------------
// Fast path: whole tiles of 64 * L frames (L = CASCADE_L = 16), CG channels (CG even) per workgroup, one wave
// per channel.
//
// Per tile the workgroup runs two phases separated by LDS-only barriers:
//   I/O     every thread owns K = L / 2 slots (frame t, channel pair cp) of the LDS tile: it reads the finished
//           results of the previous tile out of its slots, drops the prefetched inputs of this tile into the same
//           slots, sends the results to HBM (interleaved slab and / or the convolver's pair ring -- both are
//           16-byte (c, c+1) elements, i.e. exactly a slot) and issues the loads of the next tile.
//   compute each wave reads its channel's row (L samples per lane), runs all sections, writes the row back.
// Loads and stores are asynchronous across the compute phase (lds_barrier does not drain vmcnt): HBM traffic and
// the fp64 recurrences overlap.  With only two waves per SIMD (one wave per channel is all the parallelism a
// recurrence offers) the compute phase is latency-bound unless nothing waits on memory, so
//   * the wave-uniform constants of a section come through SCALAR loads into SGPRs (no LDS round trips, no VGPRs),
//     the next section's recurrence coefficients one section ahead;
//   * the second scan of the generic kernel is replaced by one 2x2 product with a per-lane matrix
//     Q[lane % 16] = P^(L (lane % 16 + 1)) (LDS table, requested before the recurrence that hides its latency);
//   * the zero-input response of a lane's incoming state (2 FMA + 1 MUL per sample, no constants but the pole
//     coefficients) is fused into the NEXT section's recurrence loop, whose dependent chain it fills.
template <int CG> struct FastCfg {
	static constexpr int L = CASCADE_L, TILE = 64 * L, NTH = 64 * CG, K = L / 2, HP = CG / 2;
	static constexpr int CHS = 64 * (L + 1) + 16 / CG;      // row stride: rows of the CG / 2 pairs on distinct bank groups
};

// zero-input response still to be added to v: r[i] = x0, (x0, x1) <- (nc3 x0 + x1, nc4 x0); all zero = nothing pending
struct PendingFix { double x0, x1, nc3, nc4; };

template <int L>
__device__ __forceinline__ void apply_fix(double (&v)[L], PendingFix f)
{
#pragma unroll
	for (int i = 0; i < L; ++i) {
		const double r = f.x0;
		v[i] += r;
		f.x0 = fma(f.nc3, r, f.x1);
		f.x1 = f.nc4 * r;
	}
}

// first 64 bytes of an op's constants: what the recurrence loop needs, fetched one op ahead
struct OpHead { long long kind; double g, c0, c1, c2, c3, c4, pad; };

__device__ __forceinline__ OpHead load_head(const double *__restrict__ od)
{
	OpHead h;
	h.kind = __double_as_longlong(od[0]); h.g = od[1]; h.c0 = od[2]; h.c1 = od[3]; h.c2 = od[4]; h.c3 = od[5]; h.c4 = od[6]; h.pad = 0.0;
	return h;
}

// one op (section j) of a channel on the tile held in v.  od: the op's [FOP_DOUBLES] constants in global memory (uniform
// address -> scalar loads), cur: their first 64 bytes (fetched by the caller, one op ahead); q: LDS [n_ops][FQ_DOUBLES];
// st: LDS [n_ops][2].  `fix` / `pending` carry the zero-input response that the NEXT op (or the caller, at the end) adds.
template <int L>
__device__ __forceinline__ void run_op_fast(double (&v)[L], long long kind, const OpHead &cur, const double *__restrict__ od, const double *q, int j, double *st,
                                            int lane, PendingFix &fix, bool &pending)
{
	const int row = lane >> 4;
	if (kind == OP_BIQUAD) {
		// requested now, consumed after the recurrence that hides their latency
		double Pw[16], P16[4];
#pragma unroll
		for (int i = 0; i < 16; ++i) Pw[i] = od[FOP_PW + i];
#pragma unroll
		for (int i = 0; i < 4; ++i) P16[i] = od[FOP_P16 + i];
		const double2 qa = *reinterpret_cast<const double2 *>(q + j * FQ_DOUBLES + 4 * (lane & 15));
		const double2 qb = *reinterpret_cast<const double2 *>(q + j * FQ_DOUBLES + 4 * (lane & 15) + 2);
		const double2 xin = *reinterpret_cast<const double2 *>(st + 2 * j);
		__builtin_amdgcn_sched_barrier(0);
		const double c0 = cur.c0, c1 = cur.c1, c2 = cur.c2, nc3 = -cur.c3, nc4 = -cur.c4;
		double m0 = 0.0, m1 = 0.0;
		{
			// zero-state recurrence of this section on (previous section's output + its pending zero-input response)
			double x0 = fix.x0, x1 = fix.x1;
#pragma unroll
			for (int i = 0; i < L; ++i) {
				const double s = v[i] + x0;
				const double t = fix.nc4 * x0;
				x0 = fma(fix.nc3, x0, x1);
				x1 = t;
				const double r = fma(c0, s, m0);
				m0 = fma(nc3, r, fma(c1, s, m1));
				m1 = fma(nc4, r, c2 * s);
				v[i] = r;
			}
		}
		__builtin_amdgcn_sched_barrier(0);
		// within-row inclusive scan of the zero-state end states; row totals in lanes 15, 31, 47
		row_scan(m0, m1, Pw);
		const double T00 = readlane_f64(m0, 15), T01 = readlane_f64(m1, 15);
		const double T10 = readlane_f64(m0, 31), T11 = readlane_f64(m1, 31);
		const double T20 = readlane_f64(m0, 47), T21 = readlane_f64(m1, 47);
		// true state at the end of each row: E_(-1) = carried state, E_r = P^(16L) E_(r-1) + T_r
		const double E00 = fma(P16[0], xin.x, fma(P16[1], xin.y, T00)), E01 = fma(P16[2], xin.x, fma(P16[3], xin.y, T01));
		const double E10 = fma(P16[0], E00, fma(P16[1], E01, T10)), E11 = fma(P16[2], E00, fma(P16[3], E01, T11));
		const double E20 = fma(P16[0], E10, fma(P16[1], E11, T20)), E21 = fma(P16[2], E10, fma(P16[3], E11, T21));
		const double cr0 = (row == 0) ? xin.x : (row == 1) ? E00 : (row == 2) ? E10 : E20;
		const double cr1 = (row == 0) ? xin.y : (row == 1) ? E01 : (row == 2) ? E11 : E21;
		m0 = fma(qa.x, cr0, fma(qa.y, cr1, m0));        // true state after this lane's samples
		m1 = fma(qb.x, cr0, fma(qb.y, cr1, m1));
		double x0 = dpp_f64<DPP_WAVE_SHR1>(m0), x1 = dpp_f64<DPP_WAVE_SHR1>(m1);
		if (lane == 0) { x0 = xin.x; x1 = xin.y; }
		fix.x0 = x0; fix.x1 = x1; fix.nc3 = nc3; fix.nc4 = nc4;
		pending = true;
		// state after the last lane's samples = the reference's (m0, m1) at that point
		if (lane == 63) *reinterpret_cast<double2 *>(st + 2 * j) = make_double2(m0, m1);
	}
	else if (kind == OP_MUL || kind == OP_ADD) {
		if (pending) apply_fix<L>(v, fix);
		fix.x0 = 0.0; fix.x1 = 0.0; fix.nc3 = 0.0; fix.nc4 = 0.0;
		pending = false;
		const double g = cur.g;
		if (kind == OP_MUL) {
#pragma unroll
			for (int i = 0; i < L; ++i) v[i] = __dmul_rn(v[i], g);
		}
		else {
#pragma unroll
			for (int i = 0; i < L; ++i) v[i] = __dadd_rn(v[i], g);
		}
	}
}

// cf: this channel's [n_ops][FOP_DOUBLES] constants; the ops [j_lo, j_hi) are run
template <int L>
__device__ __forceinline__ void run_ops_fast(double (&v)[L], const double *__restrict__ cf, const double *q, int j_lo, int j_hi, double *st, int lane)
{
	PendingFix fix = { 0.0, 0.0, 0.0, 0.0 };
	bool pending = false;            // wave-uniform: gain / add on their own stay single IEEE operations (no `+ 0.0`: keeps -0.0)
	OpHead cur = load_head(cf + j_lo * FOP_DOUBLES);
	for (int j = j_lo; j < j_hi; ++j) {
		// the kind is looked at BEFORE the next head is requested: s_waitcnt cannot tell scalar loads apart, and a wait for
		// `cur` behind that request would expose a scalar-load round trip in every op
		const long long kind = cur.kind;
		__builtin_amdgcn_sched_barrier(0);
		const OpHead nxt = load_head(cf + ((j + 1 < j_hi) ? j + 1 : j) * FOP_DOUBLES);      // in flight during this op
		run_op_fast<L>(v, kind, cur, cf + j * FOP_DOUBLES, q, j, st, lane, fix, pending);
		cur = nxt;
	}
	if (pending) apply_fix<L>(v, fix);
}

template <int CG>
__global__ __launch_bounds__(64 * CG) void cascade_fast(CascadeParams p, const double *__restrict__ fops)
{
	using Cfg = FastCfg<CG>;
	constexpr int L = Cfg::L, TILE = Cfg::TILE, NTH = Cfg::NTH, K = Cfg::K, HP = Cfg::HP, CHS = Cfg::CHS;
	extern __shared__ __attribute__((aligned(16))) double smem[];
	int s, grp;
	stream_and_group_of_block(p.xcd_map, s, grp);
	const int c0 = p.cg0 + grp * CG;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	double *tile = smem;                                        // [CG][CHS]
	double *st = tile + (size_t) CG * CHS;                      // [CG][n_ops][2]
	double *qt = st + (size_t) CG * p.n_ops * 2;                // [CG][n_ops][FQ_DOUBLES]

	const int n_st = CG * p.n_ops * 2;
	double *gstate = p.state + ((size_t) s * p.C + c0) * p.n_ops * 2;
	for (int i = tid; i < n_st; i += NTH) st[i] = gstate[i];
	for (int i = tid; i < CG * p.n_ops * FQ_DOUBLES; i += NTH) qt[i] = p.fq[(size_t) c0 * p.n_ops * FQ_DOUBLES + i];

	const long n_full = p.frames / TILE;
	const double *in = p.in + (size_t) s * p.in_stride_frames * p.C + c0;
	double *out = p.out + (size_t) s * p.out_stride_frames * p.C + c0;
	// slot k of this thread: frame t_k = (tid + k NTH) / HP, pair cp (HP divides NTH: one cp per thread)
	const int cp = tid % HP;
	double2 *ring = p.ring.base ? reinterpret_cast<double2 *>(p.ring.base) + ((size_t) s * p.ring.rows_per_stream + (c0 >> 1) + cp) * p.ring.row_stride : nullptr;
	int tk[K], la[K];
#pragma unroll
	for (int k = 0; k < K; ++k) {
		tk[k] = (tid + k * NTH) / HP;
		la[k] = (2 * cp) * CHS + tk[k] + tk[k] / L;
	}
	// ring-only output: results leave in row order instead -- slot (pair (tid + k NTH) / TILE, frame (tid + k NTH) % TILE),
	// so that one store instruction of the workgroup covers NTH consecutive 16-byte elements of ONE ring row (the
	// (t, cp) order above would interleave the rows of the HP pairs lane by lane).  Costs one extra LDS barrier.
	const bool ring_rows = ring && !p.write_interleaved;
	double2 *ring0 = p.ring.base ? reinterpret_cast<double2 *>(p.ring.base) + ((size_t) s * p.ring.rows_per_stream + (c0 >> 1)) * p.ring.row_stride : nullptr;
	int lb[K];
#pragma unroll
	for (int k = 0; k < K; ++k) {
		const int e = tid + k * NTH, tb = e % TILE;
		lb[k] = (2 * (e / TILE)) * CHS + tb + tb / L;
	}
	double2 pf0[K];
#pragma unroll
	for (int k = 0; k < K; ++k) pf0[k] = *reinterpret_cast<const double2 *>(in + (size_t) tk[k] * p.C + 2 * cp);
	const double *__restrict__ cf = fops + (size_t) (c0 + wave) * p.n_ops * FOP_DOUBLES;
	const double *wq = qt + (size_t) wave * p.n_ops * FQ_DOUBLES;
	double *row = tile + wave * CHS;
	double *cst = st + wave * p.n_ops * 2;
	__syncthreads();

	// one tile step: I/O phase (results of tile tl-1 out, inputs of tile tl in from PF, loads of tile tl+1 into PF),
	// then the compute phase on tile tl.  (Two tiles of loads in flight were measured: no gain, 32 more VGPRs.)
#define CASCADE_FAST_STEP(PF)                                                                                              \
	{                                                                                                                      \
		double2 r[K];                                                                                                      \
		if (tl > 0) {                                                                                                      \
			if (ring_rows) {                                                                                               \
				_Pragma("unroll") for (int k = 0; k < K; ++k) r[k] = make_double2(tile[lb[k]], tile[lb[k] + CHS]);        \
				lds_barrier();   /* other threads' slots were read: they may be overwritten now */                        \
			}                                                                                                              \
			else {                                                                                                         \
				_Pragma("unroll") for (int k = 0; k < K; ++k) r[k] = make_double2(tile[la[k]], tile[la[k] + CHS]);        \
			}                                                                                                              \
		}                                                                                                                  \
		if (tl < n_full) {                                                                                                 \
			_Pragma("unroll") for (int k = 0; k < K; ++k) { tile[la[k]] = PF[k].x; tile[la[k] + CHS] = PF[k].y; }        \
		}                                                                                                                  \
		if (tl > 0) {                                                                                    \
			const long t0 = (tl - 1) * TILE;                                                                               \
			if (p.write_interleaved) {                                                                                     \
				_Pragma("unroll") for (int k = 0; k < K; ++k)                                                             \
					*reinterpret_cast<double2 *>(out + (size_t) (t0 + tk[k]) * p.C + 2 * cp) = r[k];                      \
			}                                                                                                              \
			if (ring_rows) {                                                                                               \
				_Pragma("unroll") for (int k = 0; k < K; ++k)                                                             \
					ring0[(size_t) ((tid + k * NTH) / TILE) * p.ring.row_stride + ((p.ring.pos + t0 + (tid + k * NTH) % TILE) & p.ring.mask)] = r[k]; \
			}                                                                                                              \
			else if (ring) {                                                                                               \
				_Pragma("unroll") for (int k = 0; k < K; ++k) ring[(p.ring.pos + t0 + tk[k]) & p.ring.mask] = r[k];      \
			}                                                                                                              \
		}                                                                                                                  \
		if (tl + 1 < n_full) {                                                                           \
			const double *nx = in + (size_t) (tl + 1) * TILE * p.C;                                                        \
			_Pragma("unroll") for (int k = 0; k < K; ++k)                                                                 \
				PF[k] = *reinterpret_cast<const double2 *>(nx + (size_t) tk[k] * p.C + 2 * cp);                           \
		}                                                                                                                  \
		if (tl == n_full) break;                                                                                           \
		lds_barrier();                                                                                                     \
		{                                                                                                                  \
			double v[L];                                                                                                   \
			_Pragma("unroll") for (int i = 0; i < L; ++i) v[i] = row[lane * (L + 1) + i];                                 \
			run_ops_fast<L>(v, cf, wq, 0, p.n_ops, cst, lane);                                                             \
			_Pragma("unroll") for (int i = 0; i < L; ++i) row[lane * (L + 1) + i] = v[i];                                 \
		}                                                                                                                  \
		lds_barrier();                                                                                                     \
	}
	for (long tl = 0; ; ++tl) {
		CASCADE_FAST_STEP(pf0)
	}
#undef CASCADE_FAST_STEP
	for (int i = tid; i < n_st; i += NTH) gstate[i] = st[i];
}

template <int CG> static size_t fast_lds_bytes(int n_ops)
{
	using Cfg = FastCfg<CG>;
	return ((size_t) CG * Cfg::CHS + (size_t) CG * n_ops * 2 + (size_t) CG * n_ops * FQ_DOUBLES) * sizeof(double);
}

template <int CG> static bool try_launch_fast(const CascadeParams &p, int n_streams, hipStream_t stream)
{
	const size_t lds = fast_lds_bytes<CG>(p.n_ops);
	if (lds > 160 * 1024) return false;
	grant_dynamic_lds(reinterpret_cast<const void *>(cascade_fast<CG>), lds);
	dim3 grid(n_streams, p.C / CG), block(64 * CG);
	hipLaunchKernelGGL((cascade_fast<CG>), grid, block, lds, stream, p, p.fops);
	return true;
}

// 0 = not eligible; otherwise the number of leading frames the fast kernel took
static long launch_cascade_fast(const CascadeParams &p, int n_streams, hipStream_t stream)
{
	static const int env_cg = [] { const char *e = getenv("DSP_AMD_CASCADE_FAST"); return e ? atoi(e) : -1; }();   // channels per workgroup (8, 4, 2) or 0 to disable
	// measured on MI355X (256 x 8 ch, 10 sections): whole frames per workgroup (8 channels) win when the interleaved
	// slab is written (2.4 vs 3.8 ms); half frames (two independent workgroups per stream) win when only the
	// convolver's ring rows are written (2.45 vs 2.57 ms)
	// ... from about 8 sections on; with fewer the launch is memory-bound and whole frames win again (1 section: 1.25 vs 1.84 ms)
	const int cfg_cg = (env_cg >= 0) ? env_cg : ((p.ring.base && !p.write_interleaved && p.n_ops >= 8) ? 4 : 8);
	if (cfg_cg == 0 || (p.C & 1) || p.cg0 != 0 || !p.fops) return 0;
	if ((((size_t) p.in) | ((size_t) p.out)) & 15) return 0;
	if (p.ring.base && !p.ring.consecutive_pairs) return 0;
	const long n_full = p.frames / CASCADE_TILE;
	if (n_full < 1) return 0;
	// the widest channel group that divides C and whose tables fit in LDS
	for (int cg = cfg_cg; cg >= 2; cg >>= 1) {
		if (p.C % cg) continue;
		bool ok = false;
		if (cg == 8) ok = try_launch_fast<8>(p, n_streams, stream);
		else if (cg == 4) ok = try_launch_fast<4>(p, n_streams, stream);
		else if (cg == 2) ok = try_launch_fast<2>(p, n_streams, stream);
		if (ok) return n_full * CASCADE_TILE;
	}
	return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Rows variant: a wave = FOUR channels, one per 16-lane DPP row, lane = ROWS_L = 32 consecutive frames.
//
// In cascade_fast the cross-lane part of a section (64-lane carry: row scan, three row carries through readlanes, the
// Q product, about 90 instructions with a long dependent chain) costs as much time as the 128-instruction recurrence it
// follows (scripts/ubench/recbench: 260 of 540 ns per section and tile).  Giving every channel one DPP row instead of a
// whole wave removes the cross-row carries altogether -- the row's incoming state enters at its first lane BEFORE the
// 4-step row scan, which then spreads it -- and twice as many frames per lane halve what is left per sample:
// (256 + ~50) instructions per 4 x 512 samples against 4 x (128 + ~90) per 4 x 1024 (scripts/ubench/secbench: 304 ns of
// SIMD time per 1024 samples against 540).  Four channels per wave leave a quarter of the waves, so the time axis is
// shared between skewed waves.  The constants stay wave-uniform scalars: the four channels of a group must run
// identical biquad sections (gains among them are folded into the sections by the host: frows table, engine.cpp);
// anything else goes to cascade_fast.
constexpr int RW_L = ROWS_L, RW_ROW = 16 * (RW_L + 1);
// value of lane 15 of row 0 / 2 in every lane of row 1 / 3; rows 0 and 2 read 0.0 (two v_readlane pairs and a select)
__device__ __forceinline__ double bcast15_f64(double v, int lane)
{
	const double a = readlane_f64(v, 15), b = readlane_f64(v, 47);
	return (lane & 16) ? ((lane & 32) ? b : a) : 0.0;
}

// One section on the tile in v.  G channels per wave: 64 / G lanes each (G = 4: one DPP row, G = 2: two rows), lane = L
// consecutive frames.  pos: this lane's position inside its channel (0 .. 64 / G - 1); qrow: LDS [16][4] doubles,
// Q[i] = P^(L (i + 1)) of this section (G = 2 only: carry of the lower row's end state into lane i of the upper row).
template <int L, int G>
__device__ __forceinline__ void run_op_rows(double (&v)[L], const OpHead &cur, const double *__restrict__ od, double *st_row, const double *qrow, int j, int pos,
                                            int lane, PendingFix &fix, bool &pending)
{
	// (a second op kind with its own code in this loop makes the register allocator keep two copies of the tile: 37 spilled
	// VGPRs -- which is why gains are folded into the sections on the host)
	{                                      // (the caller skips gains that the table folded into a neighbouring section: no-op steps)
		double Pw[16];
#pragma unroll
		for (int i = 0; i < 16; ++i) Pw[i] = od[FOP_PW + i];                // P^(L 2^k), k = 0..3: requested now, used after the recurrence
		const double2 xin = *reinterpret_cast<const double2 *>(st_row + 2 * j);   // the channel's carried state
		__builtin_amdgcn_sched_barrier(0);
		const double c0 = cur.c0, c1 = cur.c1, c2 = cur.c2, nc3 = -cur.c3, nc4 = -cur.c4;
		double m0 = 0.0, m1 = 0.0;
		{
			double x0 = fix.x0, x1 = fix.x1;
#pragma unroll
			for (int i = 0; i < L; ++i) {
				const double s = v[i] + x0;
				const double t = fix.nc4 * x0;
				x0 = fma(fix.nc3, x0, x1);
				x1 = t;
				// three-address form by hand: left to itself the compiler accumulates r into m0's register (v_fmac) and then
				// copies it to the tile register -- one v_mov_b64 per sample, 10 % of the section's instructions
				double r;
				asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "s"(c0), "v"(s), "v"(m0));
				m0 = fma(nc3, r, fma(c1, s, m1));
				m1 = fma(nc4, r, c2 * s);
				v[i] = r;
				if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // bounds the scheduler's look-ahead (register pressure)
			}
		}
		__builtin_amdgcn_sched_barrier(0);
		// the carried state enters at the channel's first lane as P^L xin; the scan spreads it: afterwards (m0, m1) is the TRUE
		// state at the end of every lane's frames
		const double e0 = fma(Pw[0], xin.x, fma(Pw[1], xin.y, m0)), e1 = fma(Pw[2], xin.x, fma(Pw[3], xin.y, m1));
		if (pos == 0) { m0 = e0; m1 = e1; }
		row_scan(m0, m1, Pw);
		double lo0 = 0.0, lo1 = 0.0;
		if (G <= 2) {
			// rows above the channel's first: + Q[i] (end state of the row below, which is final by then).  Q is fetched here, not
			// in front of the recurrence: 8 VGPRs less across it
			const double2 qa = *reinterpret_cast<const double2 *>(qrow + 4 * (pos & 15));
			const double2 qb = *reinterpret_cast<const double2 *>(qrow + 4 * (pos & 15) + 2);
			if (G == 2) {
				lo0 = bcast15_f64(m0, lane); lo1 = bcast15_f64(m1, lane);           // 0.0 in the lower rows
				m0 = fma(qa.x, lo0, fma(qa.y, lo1, m0));
				m1 = fma(qb.x, lo0, fma(qb.y, lo1, m1));
			}
			else {
				// one channel per wave: rows 1, 2, 3 in turn, each taking the finished end state of the row below it
#pragma unroll
				for (int r = 1; r < 4; ++r) {
					const double e0 = readlane_f64(m0, 16 * r - 1), e1 = readlane_f64(m1, 16 * r - 1);
					if ((lane >> 4) == r) {
						lo0 = e0; lo1 = e1;
						m0 = fma(qa.x, e0, fma(qa.y, e1, m0));
						m1 = fma(qb.x, e0, fma(qb.y, e1, m1));
					}
				}
			}
		}
		// incoming state of a lane = true state of the lane before it
		double x0 = dpp_f64<DPP_ROW_SHR1>(m0), x1 = dpp_f64<DPP_ROW_SHR1>(m1);
		if (G <= 2 && (pos & 15) == 0 && pos > 0) { x0 = lo0; x1 = lo1; }
		if (pos == 0) { x0 = xin.x; x1 = xin.y; }
		fix.x0 = x0; fix.x1 = x1; fix.nc3 = nc3; fix.nc4 = nc4;
		pending = true;
		if (pos == 64 / G - 1) *reinterpret_cast<double2 *>(st_row + 2 * j) = make_double2(m0, m1);
	}
}

typedef unsigned int rw_u32x4 __attribute__((ext_vector_type(4)));
// 128-bit buffer store with the slot offset folded into the per-lane offset (soffset = 0).  With a wave-uniform SGPR soffset
// the compiler stages the four data dwords of consecutive slots in ONE register quad (v_mov right behind the store) and
// inserts no wait state -- the documented exception of the "VMEM store data > 64 bits overwritten by a VALU write" hazard.
// On gfx950 the exception does not hold: with two waves on the SIMD the first dword (low half of a double) was sporadically
// replaced by the NEXT slot's in lanes 12-15 of every row -- relative errors of 1e-7 in single output samples, different from
// run to run (scripts/check_determinism.py compares identical runs bit for bit; one wait state behind the store cures it).
// Without an SGPR soffset the compiler sees the hazard and places the wait states itself.
__device__ __forceinline__ void rw_store_b128(rw_u32x4 data, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff)
{
	__builtin_amdgcn_raw_buffer_store_b128(data, rsrc, voff + soff, 0, 0);
}
__device__ __forceinline__ double2 rw_as_d2(rw_u32x4 v) { return __builtin_bit_cast(double2, v); }
__device__ __forceinline__ rw_u32x4 rw_as_u4(double2 v) { return __builtin_bit_cast(rw_u32x4, v); }

// cascade_rows<G>: workgroup = (stream, group of G channels) x P waves.  The P waves share the TIME axis of the group the way
// round 2's cascade_wave did (wave w owns the tiles w, w + P, ..., one section per step, one LDS barrier per step, the 16-byte
// section states of the channels travel from wave to wave through LDS), so that 2048 channels still give 2048 waves.
// G = 4: one DPP row and 512 frames per channel and tile -- 1024 channels and more.  G = 2: two rows and 1024 frames per
// channel (one extra carry step from the lower to the upper row, per-lane matrices from an LDS table) -- twice the
// workgroups when the channels are few (strong scaling).
// Every wave moves its own tiles: 16-byte (frame, channel pair) elements between HBM and registers (buffer instructions:
// descriptor base in SGPRs + ONE per-lane offset register + a wave-uniform slot offset -- plain pointers cost a 64-bit
// address pair per slot and direction, 96 VGPRs), transposed to the lane-major layout of the recurrence through a
// wave-private LDS tile (rows placed so that both access patterns are bank-conflict free), one tile per n_ops steps,
// prefetched a whole tile period ahead.
constexpr int rw_tb_doubles(int L) { return 4 * 16 * (L + 1) + 16; }     // doubles per wave-private transposer (either G)
constexpr int RW_TB = rw_tb_doubles(RW_L);
// G = 4: rows of 16 x 33 doubles at 0, 528, 1072, 1600;  G = 2: rows of 32 x 33 doubles at 0, 1056  (L = 32)
template <int G, int ROW = RW_ROW> __device__ __forceinline__ int rw_row_base(int r) { return (G == 4) ? r * ROW + ((r >> 1) << 4) : r * 2 * ROW; }

// WIRE (G = 4, 2; bit 0: input, bit 1: output): the instances that also speak the wire formats -- p.in_fmt samples converted in the tile loads (read_buf_<fmt>),
// and / or the sink of dsp.c:685-699 (dither, clip, write_buf_<fmt>) applied in the tile stores (p.sink).  The plain fp64
// instance stays as it is.
// LL / WPE: frames per lane and waves per SIMD.  16 frames per lane (half the tile registers -- 127 VGPRs -- and half the
// transposer, constants P^(16 2^k)) at FOUR waves per SIMD was built and measured in round 2: 10.15 against 9.57 ms at the
// headline shape (8 waves per group; 11.1 with 4, 12.8 at three waves per SIMD with 6) -- the scan is paid per lane, so halving
// the frames per lane adds 20 % instructions, more than the occupancy gives back.  32 it is.
template <int G, int WIRE = 0, int LL = RW_L, int WPE = 2>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void cascade_rows(CascadeParams p, const double *__restrict__ frows, const double *__restrict__ frq, int P, int p2p)
{
	double sink_peak = 0.0;                                     // statistics of the sink (WIRE with p.sink.on)
	unsigned long long sink_clipped = 0;
	constexpr int L = LL, LPC = 64 / G, TILE = LPC * L, K = L / 2, KH = K / 2;   // K slots (16 B) per lane and tile (L = 32: 2048 samples)
	constexpr int FPS = (G == 4) ? 32 : 64;                            // frames covered by one slot instruction of the wave
	constexpr int RW_ROW = 16 * (L + 1), RW_TB = rw_tb_doubles(L);     // (shadow the L = 32 constants of the file)
	constexpr int PARTNER = (G == 4) ? RW_ROW : 2 * RW_ROW;            // LDS distance between the two channels of a pair
	// the tile traffic of a wave is one step of its sequence (see the main loop).  (Two steps for G = 1 -- results out in one,
	// the new tile in in the next -- were measured: 0.452 against 0.432 ms at 32 streams.  What G = 1 pays for is not the length
	// of that step but the 8-byte accesses themselves: each of the 8 channel groups of a stream touches every 64-byte frame.)
	constexpr int IO_STEPS = 1;
	// G = 1: one channel per wave -- 8-byte elements (32 per lane and tile), frame lane + 64 k at LDS lane + lane / 32 + 66 k
	extern __shared__ __attribute__((aligned(16))) double smem[];
	int s, grp;
	stream_and_group_of_block(p.xcd_map, s, grp);
	const int c0 = p.cg0 + grp * G;
	const int tid = threadIdx.x, lane = tid & 63, nth = 64 * P;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // position in the wavefront
	const int n_ops = p.n_ops;
	double *st = smem;                                          // [G][n_ops][2]
	double *qt = st + G * n_ops * 2;                            // [n_ops][16][4] (G <= 2 only)
	double *tb = qt + ((G <= 2) ? n_ops * FQ_DOUBLES : 0) + (size_t) w * RW_TB;       // this wave's transposer
	// point-to-point ordering instead of a workgroup barrier per step (p2p): prog[v] = section steps wave v has completed.
	// Section j of tile t only needs section j of tile t - 1 -- the neighbouring wave's work of one step earlier -- so a wave
	// waits for exactly that, and a long step of one wave (its tile traffic) no longer stops the other P - 1: with a barrier
	// per step a tile costs 7 section steps + 4 steps as long as the slowest wave's, without 10 + 1 of its own
	volatile int *prog = reinterpret_cast<volatile int *>(qt + ((G <= 2) ? n_ops * FQ_DOUBLES : 0) + (size_t) P * RW_TB);
	if (tid < 16) prog[tid] = 0;

	const int n_st = G * n_ops * 2;
	double *gstate = p.state + ((size_t) s * p.C + c0) * n_ops * 2;
	for (int i = tid; i < n_st; i += nth) st[i] = gstate[i];
	if (G <= 2) for (int i = tid; i < n_ops * FQ_DOUBLES; i += nth) qt[i] = frq[(size_t) (c0 >> 1) * n_ops * FQ_DOUBLES + i];

	const long n_full = p.frames / TILE;
	constexpr int RSRC_FLAGS = 0x00020000;                      // raw buffer, 32-bit offsets
	// bytes per sample either side: 8 (fp64), 4 (s24 / s32 / float) or 2 (s16)
	constexpr bool WIN = (WIRE & 1) != 0, WOUT = (WIRE & 2) != 0;
	const int in_fmt = WIN ? p.in_fmt : PCM_DOUBLE, out_fmt = WOUT ? p.sink.fmt : PCM_DOUBLE;
	const int in_bs = !WIN ? 8 : (in_fmt == PCM_DOUBLE) ? 8 : (in_fmt == PCM_S16) ? 2 : 4;
	const int out_bs = !WOUT ? 8 : (out_fmt == PCM_DOUBLE) ? 8 : (out_fmt == PCM_S16) ? 2 : 4;
	const bool sink_on = WOUT;
	const WordFormat wf_in = word_format(in_fmt), wf_out = word_format(out_fmt);
	const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(
		const_cast<char *>(reinterpret_cast<const char *>(p.in) + ((size_t) s * p.in_stride_frames * p.C + c0) * in_bs), 0, rsrc_records(((long) p.frames * p.C - c0) * in_bs), RSRC_FLAGS);
	const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(
		reinterpret_cast<char *>(p.out) + ((size_t) s * p.out_stride_frames * p.C + c0) * out_bs, 0, rsrc_records(((long) p.frames * p.C - c0) * out_bs), RSRC_FLAGS);
	const __amdgpu_buffer_rsrc_t r_ring = __builtin_amdgcn_make_buffer_rsrc(
		p.ring.base ? p.ring.base + 2 * (((size_t) s * p.ring.rows_per_stream + (c0 >> 1)) * p.ring.row_stride) + ((G == 1) ? (c0 & 1) : 0) : p.out, 0,
		// (G = 4: the rows of the group's two pairs; G = 2: one row; G = 1: one channel's half of the row's elements)
		p.ring.base ? rsrc_records(((G == 4) ? p.ring.row_stride : 0) * 16 + (p.ring.mask + 1) * 16 - ((G == 1) ? (c0 & 1) * 8 : 0)) : 0, RSRC_FLAGS);
	const bool has_ring = p.ring.base != nullptr;
	const int ring_row_bytes = (int) (p.ring.row_stride * 16);
	// slab order: slot k of a lane = (frame f0 + FPS k, pair pr).  G = 4: 64 lanes cover 32 frames x 2 pairs (32 contiguous
	// bytes per frame); G = 2: 64 frames of the one pair
	const int pr = (G == 4) ? (lane & 1) : 0, f0 = (G == 4) ? (lane >> 1) : lane;
	const int vo_slab = (f0 * p.C + 2 * pr) * 8, so_slab = FPS * p.C * 8, tile_bytes = TILE * p.C * 8;
	const int vo_in = (f0 * p.C + 2 * pr) * in_bs, so_in = FPS * p.C * in_bs, tile_bytes_in = TILE * p.C * in_bs;
	const int vo_out = (f0 * p.C + 2 * pr) * out_bs, so_out = FPS * p.C * out_bs, tile_bytes_out = TILE * p.C * out_bs;
	// LDS position of slot k: G = 4: row 2 pr, frame f0 + 32 k at + 33 k;  G = 2: frame lane + 64 k at lane + lane / 32 + 66 k
	double *tb_slab = tb + ((G == 4) ? rw_row_base<G, RW_ROW>(2 * pr) + f0 + f0 / L : lane + lane / L);   // frame f of a row at f + f / L
	constexpr int SLAB_K = FPS + FPS / L;
	// row order (ring-only output, G = 4): slot k = pair k >> 3, frame lane + 64 (k & 7) -- one store instruction = 1 KB of ONE
	// ring row.  (G = 2 has one pair: slab order is row order.)
	double *tb_rows = tb + lane + lane / L;                     // + rw_row_base(2 (k / KH)) + (64 + 64 / L) (k % KH)
	const int ch = lane / LPC, pos = lane % LPC;                // this lane's channel inside the group and position in it
	double *tb_lane = tb + ((G == 1) ? 0 : rw_row_base<G, RW_ROW>(ch)) + pos * (L + 1);
	double *st_row = st + ch * n_ops * 2;
	const double *__restrict__ cf = frows + (size_t) (c0 >> 1) * n_ops * FOP_DOUBLES;     // one entry per channel PAIR

	// the last tile's owner finishes last: wave wl after its tiles
	const int wl = (int) ((n_full - 1) % P);
	const long n_steps = wl + ((n_full - 1) / P + 1) * (n_ops + IO_STEPS);
	long steps = 0;
	OpHead cur = load_head(cf);
	__syncthreads();
	// settle the first head BEFORE any loop: s_waitcnt cannot tell scalar loads apart, so a wait for `cur` placed behind the
	// loads of the next head and of the scan matrices would expose a scalar-load round trip in every section
	if (cur.kind == OP_BIQUAD || cur.kind == OP_SKIP) {
		if (!p2p) for (int i = 0; i < w; ++i) lds_barrier();    // the skew: wave w starts at step w
		steps = w;
		const int wprev = (w + P - 1) % P;
		int done_steps = 0;                                     // section steps this wave has completed
		if (w < n_full) {
			double2 raw[K];                                         // G = 1: 32 eight-byte elements, element k in raw[k >> 1]
			double x[L];
			// the sink's state of this lane: generator values of the next sample it writes, statistics (WIRE only)
			typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
			const bool dither = sink_on && p.sink.dither_mult != 0.0;
			uint32_t u0 = 0, u1 = 0, js0 = 1, js1 = 1, jt0 = 1, jt1 = 1;
			double &peak = sink_peak;
			unsigned long long &clipped = sink_clipped;
			if constexpr (WOUT) {
				if (dither) {
					// sample n of the stream (interleaved order) uses A^(n + 1): this lane starts at slot 0 of tile w, walks the
					// slots of a tile FPS frames apart and the tiles of its wave P tiles apart
					const uint64_t n0 = (uint64_t) p.sink.samples_before + (uint64_t) (((long) w * TILE + f0) * p.C + c0 + 2 * pr);
					u0 = pm_pow<0>(n0 + 1); u1 = pm_pow<1>(n0 + 1);
					js0 = pm_pow<0>((uint64_t) FPS * p.C); js1 = pm_pow<1>((uint64_t) FPS * p.C);
					jt0 = pm_pow<0>((uint64_t) (P - 1) * TILE * p.C); jt1 = pm_pow<1>((uint64_t) (P - 1) * TILE * p.C);
				}
			}
			auto load_raw = [&](long t) {
				const int tbb = (int) t * tile_bytes;
				if constexpr (WIN && G == 1) {
					// one channel per wave: element k (frame lane + 64 k) as it comes -- 8 / 4 / 2 bytes -- in raw[k >> 1].x / .y
					typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
					const int tbi = (int) t * tile_bytes_in;
#pragma unroll
					for (int k = 0; k < 2 * K; ++k) {
						u32x2 v;
						if (in_bs == 8) v = __builtin_amdgcn_raw_buffer_load_b64(r_in, vo_in, tbi + k * so_in, 0);
						else if (in_bs == 4) v = u32x2{ __builtin_amdgcn_raw_buffer_load_b32(r_in, vo_in, tbi + k * so_in, 0), 0u };
						else v = u32x2{ (unsigned int) __builtin_amdgcn_raw_buffer_load_b16(r_in, vo_in, tbi + k * so_in, 0), 0u };
						if (k & 1) raw[k >> 1].y = __builtin_bit_cast(double, v); else raw[k >> 1].x = __builtin_bit_cast(double, v);
					}
				}
				else if constexpr (WIN) {
					// raw[k] holds the slot as it comes: 16 / 8 / 4 bytes of it
					const int tbi = (int) t * tile_bytes_in;
					if (in_bs == 8) {
#pragma unroll
						for (int k = 0; k < K; ++k) raw[k] = rw_as_d2(__builtin_amdgcn_raw_buffer_load_b128(r_in, vo_in, tbi + k * so_in, 0));
					}
					else if (in_bs == 4) {
#pragma unroll
						for (int k = 0; k < K; ++k) {
							const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r_in, vo_in, tbi + k * so_in, 0);
							raw[k].x = __builtin_bit_cast(double, v);
						}
					}
					else {
#pragma unroll
						for (int k = 0; k < K; ++k) {
							const u32x2 v = { __builtin_amdgcn_raw_buffer_load_b32(r_in, vo_in, tbi + k * so_in, 0), 0u };
							raw[k].x = __builtin_bit_cast(double, v);
						}
					}
				}
				else if constexpr (G == 1) {
#pragma unroll
					for (int k = 0; k < 2 * K; ++k) {
						const double v = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r_in, vo_slab, tbb + k * so_slab, 0));
						if (k & 1) raw[k >> 1].y = v; else raw[k >> 1].x = v;
					}
				}
				else {
#pragma unroll
					for (int k = 0; k < K; ++k) raw[k] = rw_as_d2(__builtin_amdgcn_raw_buffer_load_b128(r_in, vo_slab, tbb + k * so_slab, 0));
				}
			};
			auto raw_to_tb = [&]() {
				if constexpr (WIN && G == 1) {
					typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
					for (int k = 0; k < 2 * K; ++k) {
						const double r = (k & 1) ? raw[k >> 1].y : raw[k >> 1].x;
						const u32x2 v = __builtin_bit_cast(u32x2, r);
						tb_rows[(64 + 64 / L) * k] = (in_bs == 8) ? r : (in_bs == 4) ? pcm_from_word(v.x, wf_in) : pcm_from_s16(v.x & 0xffffu);
					}
				}
				else if constexpr (WIN) {
					// (one loop per sample size, the format of the 4-byte ones folded into constants: no branch per value)
					if (in_bs == 8) {
#pragma unroll
						for (int k = 0; k < K; ++k) { tb_slab[SLAB_K * k] = raw[k].x; tb_slab[SLAB_K * k + PARTNER] = raw[k].y; }
					}
					else if (in_bs == 4) {
#pragma unroll
						for (int k = 0; k < K; ++k) {
							const u32x2 v = __builtin_bit_cast(u32x2, raw[k].x);
							tb_slab[SLAB_K * k] = pcm_from_word(v.x, wf_in); tb_slab[SLAB_K * k + PARTNER] = pcm_from_word(v.y, wf_in);
						}
					}
					else {
#pragma unroll
						for (int k = 0; k < K; ++k) {
							const u32x2 v = __builtin_bit_cast(u32x2, raw[k].x);
							tb_slab[SLAB_K * k] = pcm_from_s16(v.x & 0xffffu); tb_slab[SLAB_K * k + PARTNER] = pcm_from_s16(v.x >> 16);
						}
					}
				}
				else if constexpr (G == 1) {
#pragma unroll
					for (int k = 0; k < 2 * K; ++k) tb_rows[(64 + 64 / L) * k] = (k & 1) ? raw[k >> 1].y : raw[k >> 1].x;
				}
				else {
#pragma unroll
					for (int k = 0; k < K; ++k) { tb_slab[SLAB_K * k] = raw[k].x; tb_slab[SLAB_K * k + PARTNER] = raw[k].y; }
				}
			};
			load_raw(w);
			// results of tile t_out (lane-major in the transposer) -> HBM
			const bool slab_order = p.write_interleaved || G == 2;
			auto fetch_out = [&](double2 (&y)[K]) {
				if constexpr (G == 1) {
#pragma unroll
					for (int k = 0; k < 2 * K; ++k) { const double v = tb_rows[(64 + 64 / L) * k]; if (k & 1) y[k >> 1].y = v; else y[k >> 1].x = v; }
				}
				else if (slab_order) {
#pragma unroll
					for (int k = 0; k < K; ++k) y[k] = make_double2(tb_slab[SLAB_K * k], tb_slab[SLAB_K * k + PARTNER]);
				}
				else {
#pragma unroll
					for (int k = 0; k < K; ++k) {
						const double *a = tb_rows + rw_row_base<G, RW_ROW>(2 * (k / KH)) + (64 + 64 / L) * (k % KH);
						y[k] = make_double2(a[0], a[PARTNER]);
					}
				}
			};
			auto store_out = [&](const double2 (&y)[K], long t_out) {
				if constexpr (WOUT && G == 1) {
					if (sink_on) {
						// one channel per wave: element k = frame lane + 64 k of the tile, 64 C samples from slot to slot
						typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
						const int tbo = (int) t_out * tile_bytes_out;
#pragma unroll
						for (int k = 0; k < 2 * K; ++k) {
							const double a = sink_sample((k & 1) ? y[k >> 1].y : y[k >> 1].x, dither, u0, u1, p.sink.dither_mult, peak, clipped);
							if (dither) { u0 = pm_mul(u0, js0); u1 = pm_mul(u1, js1); }
							if (out_bs == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a), r_out, vo_out + tbo + k * so_out, 0, 0);
							else if (out_bs == 4) __builtin_amdgcn_raw_buffer_store_b32(pcm_to_word(a, wf_out), r_out, vo_out + tbo + k * so_out, 0, 0);
							else __builtin_amdgcn_raw_buffer_store_b16((unsigned short) pcm_to_s16(a), r_out, vo_out + tbo + k * so_out, 0, 0);
						}
						if (dither) { u0 = pm_mul(u0, jt0); u1 = pm_mul(u1, jt1); }   // 32 slots = one tile on: the wave's next tile is P - 1 further
						return;
					}
				}
				else if constexpr (WOUT) {
					if (sink_on) {
						// the last kernel of the pipeline: dither, clip and convert on the way out (slab order; no ring behind a sink)
						const int tbo = (int) t_out * tile_bytes_out;
#pragma unroll
						for (int k = 0; k < K; ++k) {
							uint32_t b0 = 0, b1 = 0;
							if (dither) { b0 = pm_mul(u0, PM_A0); b1 = pm_mul(u1, PM_A1); }
							const double a = sink_sample(y[k].x, dither, u0, u1, p.sink.dither_mult, peak, clipped);
							const double b = sink_sample(y[k].y, dither, b0, b1, p.sink.dither_mult, peak, clipped);
							if (dither) { u0 = pm_mul(u0, js0); u1 = pm_mul(u1, js1); }
							if (out_bs == 8) rw_store_b128(rw_as_u4(make_double2(a, b)), r_out, vo_out, tbo + k * so_out);
							else if (out_bs == 4) {
								const u32x2 v = { pcm_to_word(a, wf_out), pcm_to_word(b, wf_out) };
								__builtin_amdgcn_raw_buffer_store_b64(v, r_out, vo_out + tbo + k * so_out, 0, 0);
							}
							else __builtin_amdgcn_raw_buffer_store_b32(pcm_to_s16(a) | (pcm_to_s16(b) << 16), r_out, vo_out + tbo + k * so_out, 0, 0);
						}
						if (dither) { u0 = pm_mul(u0, jt0); u1 = pm_mul(u1, jt1); }   // 16 slots = one tile on: the wave's next tile is P - 1 further
						return;
					}
				}
				if constexpr (G == 1) {
					typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
					const int tbb = (int) t_out * tile_bytes;
					const long e0 = p.ring.pos + t_out * TILE + lane;
#pragma unroll
					for (int k = 0; k < 2 * K; ++k) {
						const u32x2 v = __builtin_bit_cast(u32x2, (k & 1) ? y[k >> 1].y : y[k >> 1].x);
						if (p.write_interleaved) __builtin_amdgcn_raw_buffer_store_b64(v, r_out, vo_slab + tbb + k * so_slab, 0, 0);
						if (has_ring) __builtin_amdgcn_raw_buffer_store_b64(v, r_ring, (int) ((e0 + 64 * k) & p.ring.mask) * 16, 0, 0);
					}
					return;
				}
				if (p.write_interleaved) {
					const int tbb = (int) t_out * tile_bytes;
#pragma unroll
					for (int k = 0; k < K; ++k) rw_store_b128(rw_as_u4(y[k]), r_out, vo_slab, tbb + k * so_slab);
				}
				if (has_ring && slab_order) {
					const long e0 = p.ring.pos + t_out * TILE + f0;
#pragma unroll
					for (int k = 0; k < K; ++k)
						rw_store_b128(rw_as_u4(y[k]), r_ring, (int) ((e0 + FPS * k) & p.ring.mask) * 16 + pr * ring_row_bytes, 0);
				}
				else if (has_ring) {
					const long e0 = p.ring.pos + t_out * TILE + lane;
#pragma unroll
					for (int k = 0; k < K; ++k)
						rw_store_b128(rw_as_u4(y[k]), r_ring, (int) ((e0 + 64 * (k % KH)) & p.ring.mask) * 16, (k / KH) * ring_row_bytes);
				}
			};
			for (long t = w; t < n_full; t += P) {
				// Order matters (vmcnt counts loads and stores together, in order): the previous tile's results are read out of the
				// transposer, THEN this tile's loads are waited for (nothing else is in flight: the stores before them went out a
				// whole tile ago), and only then do the stores and the next tile's loads go out.
				double2 y[K];
				if (t > w) {
#pragma unroll
					for (int i = 0; i < L; ++i) tb_lane[i] = x[i];
					fetch_out(y);
				}
				raw_to_tb();
				if (t > w) store_out(y, t - P);
#pragma unroll
				for (int i = 0; i < L; ++i) x[i] = tb_lane[i];
				// the tile is settled HERE, once: left in flight into the section loop it makes the compiler wait for all LDS and
				// scalar traffic (one counter) in front of every recurrence -- behind the requests that the recurrence should hide
#pragma unroll
				for (int i = 0; i < L; ++i) asm volatile("" : "+v"(x[i]));
				PendingFix fix = { 0.0, 0.0, 0.0, 0.0 };
				bool pending = false;
				auto step = [&](int j) {
					// the kind is looked at BEFORE the next head is requested: a wait for `cur` behind that request (s_waitcnt cannot
					// tell scalar loads apart) would expose a scalar-load round trip in every step
					const bool section = (cur.kind == OP_BIQUAD);
					if (p2p) {
						// the predecessor's section j of the tile before this one: wave w - 1's step of the same index, or -- for
						// wave 0 -- wave P - 1's step of one tile period earlier
						const int need = done_steps + 1 - ((w == 0) ? n_ops : 0);
						while (__builtin_amdgcn_readfirstlane(prog[wprev]) < need) __builtin_amdgcn_s_sleep(1);
						__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
					}
					__builtin_amdgcn_sched_barrier(0);
					const OpHead nxt = load_head(cf + ((j + 1 < n_ops) ? j + 1 : 0) * FOP_DOUBLES);     // in flight during this op
					if (section) run_op_rows<L, G>(x, cur, cf + j * FOP_DOUBLES, st_row, qt + j * FQ_DOUBLES, j, pos, lane, fix, pending);
					else asm volatile("" :: "s"(nxt.kind), "s"(nxt.g), "s"(nxt.c0), "s"(nxt.c1), "s"(nxt.c2), "s"(nxt.c3), "s"(nxt.c4));
					// (the no-op path settles the next head too: otherwise `cur` counts as possibly in flight at the loop header and the
					// compiler waits for ALL scalar loads in front of every recurrence)
					const double post = cur.g;                           // last op's entry: product of the gains behind the last section
					cur = nxt;
					if (j + 1 == n_ops) {
						if (pending) apply_fix<L>(x, fix);
						if (post != 1.0) {
#pragma unroll
							for (int i = 0; i < L; ++i) x[i] *= post;
						}
					}
					if (p2p) {
						++done_steps;
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");     // the section state is in LDS before the count
						if (lane == 0) prog[w] = done_steps;
					}
					else lds_barrier();       // (two sections per barrier were measured: no gain)
				};
				// The next tile's loads: unconditional (the last one re-reads this tile) -- a conditional load has to select between
				// old and new registers, which costs a wait right behind the loads -- and at the boundary (issuing them one step
				// later, to shorten the boundary step everybody waits for, was measured: slower for G = 1 and 2, no gain for 4)
				load_raw((t + P < n_full) ? t + P : t);
				// The tile traffic is a step of its own: inside a section step it made that step (which every wave of the group
				// waits for) up to 1.8 x as long, and with P waves skewed by one step almost every step had one such wave.  As an
				// (n_ops + 1)-th step it runs beside the other waves' sections and costs one step in n_ops + 1.
				if (!p2p) lds_barrier();
				for (int j = 0; j < n_ops; ++j) step(j);
				steps += n_ops + IO_STEPS;
			}
			{
				// the last tile of this wave
				const long t_last = w + ((n_full - 1 - w) / P) * P;
				double2 y[K];
#pragma unroll
				for (int i = 0; i < L; ++i) tb_lane[i] = x[i];
				fetch_out(y);
				store_out(y, t_last);
			}
		}
		if (!p2p) for (; steps < n_steps; ++steps) lds_barrier();
	}
	__syncthreads();
	if constexpr (WOUT) { if (p.sink.stats) sink_stats_block(p.sink.stats, s, sink_peak, sink_clipped); }
	for (int i = tid; i < n_st; i += nth) gstate[i] = st[i];
}

template <int G> static long try_launch_rows(const CascadeParams &p, int n_streams, int P, hipStream_t stream)
{
	constexpr int TILE = (64 / G) * RW_L;
	const long n_full = p.frames / TILE;
	if (n_full < 1 || (p.C % G)) return 0;
	P = (int) std::min<long>(std::min(P, p.n_ops), n_full);
	const size_t lds = ((size_t) G * p.n_ops * 2 + ((G <= 2) ? (size_t) p.n_ops * FQ_DOUBLES : 0) + (size_t) P * RW_TB + 8) * sizeof(double);
	if (lds > 160 * 1024) return 0;
	// point-to-point ordering for long calls (at least 16 tiles per wave); short ones keep the workgroup barrier per step: the
	// polling costs more than it saves there (config 2's chunks of 8 tiles: 0.052 against 0.064 ms; 2048-frame calls 0.031 / 0.035)
	const int p2p = n_full / P >= 16 ? 1 : 0;
	dim3 grid(n_streams, p.C / G), block(64 * P);
	const int wire = (p.in_fmt != PCM_DOUBLE ? 1 : 0) | (p.sink.on ? 2 : 0);
	if (wire) {
		// (an instance per end: the one that only reads a wire format -- the headline's file -> file run -- does not carry the sink)
		auto go = [&](auto kernel) {
			grant_dynamic_lds(reinterpret_cast<const void *>(kernel), lds);
			hipLaunchKernelGGL(kernel, grid, block, lds, stream, p, p.frows, p.frq, P, p2p);
		};
		if (wire == 1) go(cascade_rows<G, 1>); else if (wire == 2) go(cascade_rows<G, 2>); else go(cascade_rows<G, 3>);
		return n_full * TILE;
	}
	grant_dynamic_lds(reinterpret_cast<const void *>(cascade_rows<G, 0>), lds);
	hipLaunchKernelGGL((cascade_rows<G, 0>), grid, block, lds, stream, p, p.frows, p.frq, P, p2p);
	return n_full * TILE;
}

// whether cascade_rows takes (the leading tiles of) this call, and in which shape: a pure function of the call
static bool rows_choice(const CascadeParams &p, int n_streams, int *Gout, int *Pout)
{
	static const int env = [] { const char *e = getenv("DSP_AMD_CASCADE_ROWS"); return e ? atoi(e) : -1; }();   // 0 = never, G*100 + P = force
	if (env == 0 || !p.frows || !p.frq || p.cg0 != 0) return false;
	if ((((size_t) p.in) | ((size_t) p.out)) & 15) return false;
	if (p.ring.base && !p.ring.consecutive_pairs) return false;
	// 32-bit byte offsets inside one stream's slab / ring rows
	if ((double) std::max(p.in_stride_frames, p.out_stride_frames) * p.C * 8 >= 2.0e9 || (double) p.ring.row_stride * 16 * (p.C / 2) >= 2.0e9) return false;
	// Four channels per wave when that still gives 2048 waves of at most 8 per group (what the LDS transposers allow), i.e.
	// from 1024 channels; two per wave from 512 channels; one per wave (all four DPP rows, three sequential row carries,
	// 8-byte elements) below that.  Measured, 10 sections, ms per launch at 32 / 64 / 128 / 256 streams x 8 ch: see docs/history.md section 4.1.
	const long channels = (long) n_streams * p.C;
	int G = (channels >= 1024 && p.rows4_ok) ? 4 : (channels >= 512) ? 2 : 1, P;
	const bool wire = p.in_fmt != PCM_DOUBLE || p.sink.on;
	if (env > 0) { G = env / 100; P = env % 100; if ((G == 4 && !p.rows4_ok) || (G != 4 && G != 2 && G != 1)) return false; }
	else {
		P = (int) std::min<long>(8, std::max<long>(1, (2048 * G + channels - 1) / channels));
		// with point-to-point ordering more waves per group pay even when the chip is full anyway: 8 waves in one workgroup per
		// CU against 2 x 4 (9.48 against 9.67 ms at 256 streams, scripts/exp_rowsP.sh)
		if (G == 4 && P < 8 && p.n_ops >= 8 && p.frames / 512 >= 64) P = 8;   // (long calls only: the skewed start costs P steps per launch)
	}
	if (P > 8) P = 8;
	if (P < 1) P = 1;
	const int tile = (64 / G) * RW_L;
	if (p.frames / tile < 1 || (p.C % G)) return false;
	const int Pe = (int) std::min<long>(std::min(P, p.n_ops), p.frames / tile);
	if (((size_t) G * p.n_ops * 2 + ((G <= 2) ? (size_t) p.n_ops * FQ_DOUBLES : 0) + (size_t) Pe * RW_TB + 8) * sizeof(double) > 160 * 1024) return false;
	// the wire formats are spoken by the instances that move channel pairs -- and only where the plain call would run the SAME
	// instance shape: fused or not, a call gives the same samples bit for bit
	if (wire && (!pcm_fusable(p.in_fmt) || (p.sink.on && (!pcm_fusable(p.sink.fmt) || p.ring.base || !p.write_interleaved)))) return false;
	*Gout = G; *Pout = P;
	return true;
}

bool cascade_rows_takes(const CascadeParams &p, int n_streams)
{
	int G, P;
	return rows_choice(p, n_streams, &G, &P);
}

// 0 = not eligible; otherwise the number of leading frames taken
static long launch_cascade_rows(const CascadeParams &p, int n_streams, hipStream_t stream)
{
	int G, P;
	if (!rows_choice(p, n_streams, &G, &P)) return 0;
	return (G == 4) ? try_launch_rows<4>(p, n_streams, P, stream) : (G == 2) ? try_launch_rows<2>(p, n_streams, P, stream) : try_launch_rows<1>(p, n_streams, P, stream);
}

size_t cascade_lds_bytes(int Cg, int n_ops)
{
	return ((size_t) Cg * CH_STRIDE + (size_t) Cg * n_ops * 2 + (size_t) Cg * n_ops * OPL_DOUBLES) * sizeof(double);
}

const char *launch_cascade(const CascadeParams &p_in, int n_streams, hipStream_t stream)
{
	CascadeParams p0 = p_in;
	p0.xcd_map = 1;          // (the channel groups of a stream co-scheduled on one XCD: scripts/exp_xcdmap.sh, docs/history.md section 4.1)
	CascadeParams p = p0;
	const char *name = "cascade_rows";
	// a call in wire formats: cascade_rows and the generic kernel speak them (the host asks cascade_rows_takes() first)
	const bool wire = p0.in_fmt != PCM_DOUBLE || p0.sink.on;
	long done = launch_cascade_rows(p0, n_streams, stream);
	if (done == 0 && !wire) { name = "cascade_fast"; done = launch_cascade_fast(p0, n_streams, stream); }
	if (done > 0) {
		// the generic kernel continues the streams (state is in HBM) on whatever is left of the block
		if (done == p0.frames) return name;
		p.in = reinterpret_cast<const double *>(reinterpret_cast<const char *>(p0.in) + (size_t) done * p0.C * pcm_sample_bytes(p0.in_fmt));
		p.out = reinterpret_cast<double *>(reinterpret_cast<char *>(p0.out) + (size_t) done * p0.C * (p0.sink.on ? pcm_sample_bytes(p0.sink.fmt) : 8));
		p.sink.samples_before += done * p0.C;
		p.frames = p0.frames - done;
		if (p.ring.base) p.ring.pos = (p0.ring.pos + done) & p0.ring.mask;
	}
	else name = "cascade_kernel";
	const int n_groups = (p.C - p.cg0 + p.Cg - 1) / p.Cg;
	const int waves = p.Cg < 8 ? (p.Cg < 1 ? 1 : p.Cg) : 8;
	dim3 grid(n_streams, n_groups), block(64 * waves);
	const size_t lds = cascade_lds_bytes(p.Cg, p.n_ops);
	grant_dynamic_lds(reinterpret_cast<const void *>(cascade_kernel), lds);
	hipLaunchKernelGGL(cascade_kernel, grid, block, lds, stream, p, p.ops);
	return name;
}

}  // namespace dspamd
