// kernels_resident.hip -- small plugin blocks without a launch per block (round 5; the mailbox protocol: round 6).
//
// What it serves: the reference's LADSPA frontend and CLI drive a chain with run() calls of 64 ... 1024 frames (ladspa_dsp.c:316-355, dsp.h:38); a
// kernel launch plus its completion costs 25 us on this platform whatever the block, the reference's own loop 3 us at 64 frames x 2 ch x 10 sections.
// For a device segment whose stages are remixes / weighted mixes, direct-form FIRs and at most two cascades of gains / adds / sections (the equaliser, the
// crossover, a crossover with correction FIRs, a mid/side equaliser, crossfeed: engine.cpp Pipeline::resident_plan) a single workgroup stays on the device
// for a bounded time, polls a mailbox the host writes the block into, runs the block through its passes in LDS -- a cascade as the reference's recurrence
// as written (biquad.h:76-92), sample by sample, one lane per op per channel, states and coefficients in registers; a FIR in the reference's summation
// order (fir.c:43-62); a remix as remix.c:39-101 -- and writes the output into a mailbox the host polls.
//
// The mailboxes (engine.h: ResidentUnit, ResidentParams): 16-byte units { value, request ^ bits(value) } that validate themselves, so the wave asks for the
// control unit AND the units of the block it expects in ONE burst of loads and neither side waits for an acknowledgement of its stores.  Round 5 had a
// doorbell word and plain staging buffers in host memory: a PCIe read round trip for the doorbell, a second, dependent one for the block, and a
// system-scope fence (a third round trip's worth) between the output and the completion word -- 12.1 us per 64-frame stereo block of ten sections.  The
// request mailbox now lives in device memory the CPU stores into over the BAR (page-locked host memory where it cannot): every PCIe transfer on the path is a
// posted write.
//
// Bounded lifetime: the wave leaves after `lifetime` ticks of the 100 MHz clock without a block, `max_life` ticks after its launch however busy it is kept
// (between two blocks), when the host asks it to (frames = RESIDENT_STOP), or after `max_polls` turns of its loop whatever the clock says -- so a
// hipDeviceSynchronize() anywhere in the process waits a few milliseconds at most, and no failure of the host can leave a kernel behind.  Its last store
// is alive = 0 behind a system-scope fence; the host starts another one with the next block.
// The states live in device memory between blocks (loaded and stored around every block, past the L1).  They are NOT fenced on the path of a block: when
// the wave finds nothing to do it waits for its stores and says so (ctl->settled), and before anything else touches the states (the ordinary kernels at
// another block size, reset, destroy) the host waits for that word (Resident::quiesce) -- the wave stays.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
#include "kparams.h"
#include "engine.h"

namespace dspamd {

constexpr int RES_MAX_OPS = 16;

__device__ __forceinline__ double ld_agent(const double *p)
{
	return __longlong_as_double((long long) __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(double *p, double v)
{
	__hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long) __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the value `v` of the lane below in the 16-lane row; lane 0 of a row (no lane below: the move is disabled there and the destination keeps its `old`) gets `first`
__device__ __forceinline__ double row_shr1_or(double v, double first)
{
	int lo = __double2loint(v), hi = __double2hiint(v);
	lo = __builtin_amdgcn_update_dpp(__double2loint(first), lo, 0x111, 0xf, 0xf, false);
	hi = __builtin_amdgcn_update_dpp(__double2hiint(first), hi, 0x111, 0xf, 0xf, false);
	return __hiloint2double(hi, lo);
}

typedef unsigned int res_u4 __attribute__((ext_vector_type(4)));
// sc0 | sc1 (system scope: the access goes to memory whatever a cache holds) | the compiler's volatile bit (a poll is not hoisted out of its loop)
constexpr int RES_AUX = (int) 0x80000011u;
__device__ __forceinline__ ResidentUnit ld_unit(__amdgpu_buffer_rsrc_t r, int unit)
{
	const res_u4 q = __builtin_amdgcn_raw_buffer_load_b128(r, unit * 16, 0, RES_AUX);
	ResidentUnit u;
	u.v = __hiloint2double((int) q.y, (int) q.x);
	u.w = ((unsigned long long) q.w << 32) | q.z;
	return u;
}
__device__ __forceinline__ void st_unit(__amdgpu_buffer_rsrc_t r, int unit, double v, unsigned long long rq)
{
	const unsigned long long b = (unsigned long long) __double_as_longlong(v), w = rq ^ b;
	const res_u4 q = { (unsigned) b, (unsigned) (b >> 32), (unsigned) w, (unsigned) (w >> 32) };
	__builtin_amdgcn_raw_buffer_store_b128(q, r, unit * 16, 0, 0x11);
}
__device__ __forceinline__ unsigned long long unit_request(const ResidentUnit &u) { return u.w ^ (unsigned long long) __double_as_longlong(u.v); }

constexpr int RES_SPEC = 4;                  // payload units per lane asked for together with the control unit, at most
constexpr unsigned RES_SPINS = 1u << 17;     // re-reads of a unit that does not decode before the wave gives up (the host wrote the block BEFORE the control unit)

// Workgroup = ceil(C / 4) waves; a wave = 4 channels, one per 16-lane DPP row; lane j of a row = op j of its channel (n_ops <= 16).  A block runs as a
// systolic array: at step t lane j works on frame t - j, its input the result lane j - 1 had a step earlier (one DPP move), lane 0 reads the frame from
// the block buffer, the channel's last op writes it back -- frames + n_ops - 1 steps of one dependent fma each instead of frames x n_ops of them (the
// first form of this kernel, one lane per channel: 116 us per 64-frame block of a stereo ten-section chain on a GPU that idles at a low clock).
// Every lane runs the same instructions: r = fma(a, x, b) is the section's output (a = c0, b = m0), a gain (a = g, b = -0.0: the product keeps its
// sign of zero), an add (a = 1, b = v) or a pass (a = 1, b = -0.0), bit for bit what __dmul_rn / __dadd_rn give; only sections update (m0, m1).
// GEN = false: the segment is ONE cascade (the equaliser at LADSPA block sizes: the shape the 9 us are asked of) -- no pass list, no tables, no histories
template <bool GEN>
__global__ __launch_bounds__(512) void cascade_resident(ResidentParams p)
{
	extern __shared__ __attribute__((aligned(16))) double buf[];      // two halves of the block buffer; behind them two words (the request, a failure flag), a word per lane, the FIR histories
	const int tid = threadIdx.x, nth = blockDim.x;
	const int j = tid & 15, ch = tid >> 4;                             // op and channel of this lane in a cascade pass
	const int half = p.buf_doubles / 2;
	unsigned long long *req_w = reinterpret_cast<unsigned long long *>(buf + p.buf_doubles);
	double *lane_word = buf + p.buf_doubles + 2 + tid;
	double *hist_lds = buf + p.buf_doubles + 2 + 1024;                 // [pass][channel][RES_FIR_TAPS]
	double *tab_lds = hist_lds + RES_MAX_PASSES * RES_FIR_MAX_CH * RES_FIR_TAPS;     // [pass][RES_TAB_DOUBLES]: the pass's tables, copied once (below)
	// this lane's op in each of the segment's cascades, in registers for the kernel's lifetime
	struct LaneOp { double a, b, c1, c2, c3, c4; bool biq, mine; double *stp; };
	auto lane_op = [&](int q) {
		LaneOp o{ 1.0, -0.0, 0.0, 0.0, 0.0, 0.0, false, false, nullptr };
		if (q < p.n_casc) {
			const int C = p.cs[q].C, n_ops = p.cs[q].n_ops;
			o.mine = ch < C && j < n_ops;
			o.stp = p.cs[q].state + ((size_t) ch * n_ops + j) * 2;
			if (o.mine) {
				const OpDesc &od = p.cs[q].ops[(size_t) ch * n_ops + j];
				if (od.kind == OP_BIQUAD) { o.biq = true; o.a = od.c[0]; o.c1 = od.c[1]; o.c2 = od.c[2]; o.c3 = od.c[3]; o.c4 = od.c[4]; }
				else if (od.kind == OP_MUL) o.a = od.g;
				else if (od.kind == OP_ADD) o.b = od.g;
			}
		}
		return o;
	};
	const LaneOp op0 = lane_op(0), op1 = lane_op(1);
	const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<ResidentUnit *>(p.mail_in), 0, (1 + RESIDENT_UNITS) * 16, 0x00020000);
	const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.mail_out, 0, RESIDENT_UNITS * 16, 0x00020000);
	const int spec = p.spec_units < RES_SPEC ? p.spec_units : RES_SPEC;
	unsigned done = p.done0, settled = p.done0;
	unsigned long long t_last = wall_clock64();
	const unsigned long long t_start = t_last;
	if (tid == 0) req_w[1] = 0;
	// every pass's tables into LDS, once per launch: a FIR's taps [16][32] and, behind them, its channel map [16] (ints); a remix's weights [c_out][max_n],
	// factors [c_out] and sources [c_out][max_n] (ints, -1 terminated) -- the launcher checked that they fit
	for (int k = 0; GEN && k < p.n_pass; ++k) {
		const ResidentPass &ps = p.pass[k];
		double *tb = tab_lds + k * RES_TAB_DOUBLES;
		if (ps.kind == RES_PASS_FIR) {
			for (int e = tid; e < RES_FIR_MAX_CH * RES_FIR_TAPS; e += nth) { const int f = e / RES_FIR_TAPS; bool used = false; for (int c = 0; c < ps.c_in; ++c) used |= ps.foc[c] == f; tb[e] = used ? ps.taps[e] : 0.0; }
			int *fo = reinterpret_cast<int *>(tb + RES_FIR_MAX_CH * RES_FIR_TAPS);
			for (int e = tid; e < RES_FIR_MAX_CH; e += nth) fo[e] = e < ps.c_in ? ps.foc[e] : -1;
		}
		else if (ps.kind == RES_PASS_REMIX) {
			const int cells = ps.c_out * ps.max_n;
			double *w = tb, *post = tb + cells;
			int *idx = reinterpret_cast<int *>(tb + cells + ps.c_out);
			for (int e = tid; e < cells; e += nth) { w[e] = ps.w ? ps.w[e] : 1.0; idx[e] = ps.idx[e]; }
			for (int e = tid; e < ps.c_out; e += nth) post[e] = ps.post ? ps.post[e] : 1.0;
		}
	}
	__syncthreads();
#ifdef RES_TIMING
	unsigned it_last = 0;
#endif
	for (unsigned it = 0; it < p.max_polls; ++it) {
		// one burst: the control unit and the units of the block this lane expects (offsets beyond the mailbox read as zeros and do not decode)
		const ResidentUnit cu = ld_unit(r_in, 0);
		ResidentUnit pu[RES_SPEC];
#pragma unroll
		for (int k = 0; k < RES_SPEC; ++k) if (k < spec) pu[k] = ld_unit(r_in, 1 + tid + k * nth);
		// thread 0 reads the clock and decides for everybody (the waves meet at barriers below: one decision, not one per wave)
		if (tid == 0) {
			unsigned long long rq0 = unit_request(cu);
			// nothing for a lifetime: leave.  And leave between two blocks once max_life_ticks have gone by however busy the host keeps the wave: another
			// thread's hipDeviceSynchronize() (a second chain being built while this one plays) must not wait for the audio to stop
			const unsigned long long now = wall_clock64();
			if ((unsigned) (rq0 >> 32) == done && (unsigned) (rq0 & 0xffffffffu) != RESIDENT_STOP && (now - t_last > p.lifetime_ticks || now - t_start > p.max_life_ticks))
				rq0 = ((unsigned long long) done << 32) | RESIDENT_STOP;
			req_w[0] = rq0;
		}
		__syncthreads();
		const unsigned long long rq = req_w[0];
		__syncthreads();
		const unsigned seq = (unsigned) (rq >> 32), low = (unsigned) (rq & 0xffffffffu);
		if (low == RESIDENT_STOP) break;
		if (seq == done) {
			// nothing to do: the moment to say that the last block's states are out (a wait for this wave's stores, then a word in host memory) -- the host
			// looks at it before it lets anything else touch the states, and a block never waits for it
			if (settled != done) {
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
				__syncthreads();
				if (tid == 0) __hip_atomic_store(&p.ctl->settled, done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
				settled = done;
			}
			__builtin_amdgcn_s_sleep(4);
			continue;
		}
		// ---- a block: the mailbox -> LDS, the passes, LDS -> the host's mailbox
#ifdef RES_TIMING
		const unsigned long long tt0 = wall_clock64();
#endif
		const int frames = (int) (low & RES_FRAMES_MASK);
		const int n = frames * p.Cin, n_out = frames * p.Cout;
		if (frames < 1 || n > half || n_out > half || n > RESIDENT_UNITS || n_out > RESIDENT_UNITS) break;      // (a request the host never makes: leave rather than touch anything)
		double *cur = buf, *oth = buf + half;                       // a pass that cannot work in place reads `cur`, writes `oth`, and they change places
		double m0a = 0.0, m1a = 0.0, m0b = 0.0, m1b = 0.0;               // the states of this lane's section in the first / second cascade
		if (op0.mine && op0.biq) { m0a = ld_agent(op0.stp); m1a = ld_agent(op0.stp + 1); }
		if (op1.mine && op1.biq) { m0b = ld_agent(op1.stp); m1b = ld_agent(op1.stp + 1); }
		// the histories of the FIR passes, asked for now (device memory: they arrive while the block does)
		for (int k = 0; GEN && k < p.n_pass; ++k) {
			const ResidentPass &ps = p.pass[k];
			if (ps.kind != RES_PASS_FIR) continue;
			const double *hd = ps.hist + (size_t) ((low >> (16 + k)) & 1u) * ps.c_in * RES_FIR_TAPS;
			for (int e = tid; e < ps.c_in * RES_FIR_TAPS; e += nth) hist_lds[k * RES_FIR_MAX_CH * RES_FIR_TAPS + e] = ld_agent(hd + e);
		}
		{
			// a unit that does not decode to this request has not arrived yet (or was not asked for with the control unit): read it again.  The host
			// wrote every unit of the block before the control unit, so this is the exception; a bound all the same
			bool lost = false;
			auto settle = [&](ResidentUnit u, bool have, int e) -> double {
				unsigned spins = 0;
				while (!have || unit_request(u) != rq) {
					if (++spins > RES_SPINS) { lost = true; break; }
					u = ld_unit(r_in, 1 + e);
					have = true;
				}
				return u.v;
			};
#pragma unroll
			for (int k = 0; k < RES_SPEC; ++k) { const int e = tid + k * nth; if (e < n) cur[e] = settle(pu[k], k < spec, e); }
			for (int base = RES_SPEC * nth; base < n; base += 4 * nth) {
				ResidentUnit v[4];
#pragma unroll
				for (int k = 0; k < 4; ++k) { const int e = base + k * nth + tid; if (e < n) v[k] = ld_unit(r_in, 1 + e); }
#pragma unroll
				for (int k = 0; k < 4; ++k) { const int e = base + k * nth + tid; if (e < n) cur[e] = settle(v[k], true, e); }
			}
			if (lost) req_w[1] = 1;
		}
		__syncthreads();
#ifdef RES_TIMING
		const unsigned long long tt1 = wall_clock64();
#endif
		if (req_w[1]) break;                                         // (the host times out and takes the block through a launch)
		auto cascade = [&](const LaneOp &o, const int C, const int n_ops, double m0, double m1) {
			const bool mine = o.mine, biq = o.biq;
			const double a = o.a, b = o.b, c1 = o.c1, c2 = o.c2, c3 = o.c3, c4 = o.c4;
			if (ch >= C) return;
			{
			const int nf = frames, steps = nf + n_ops - 1;
			const bool upd = mine && biq, wr = mine && j == n_ops - 1;
			const double *rd = cur + ch;                             // frame t of this row's channel at rd[t C]
			auto frame_in = [&](int t) -> double { return rd[t * C]; };
			double s0 = biq ? m0 : b;                                // the addend of r = fma(a, x, s0): a section's m0, or the op's constant
			// the channel's last op writes frame t - j at step t; every other lane writes into a word of its own behind the block (no branch around the store)
			double *wr_base = wr ? cur + ch - j * C : lane_word;
			const int wr_stride = wr ? C : 0;
			double prev = 0.0;
			// One step.  The wave is alone on its SIMD: a step costs what it ISSUES (round 6 measured 91 ns per step for round 5's 17 vector and 10 scalar
			// instructions -- 6.7 of the 10.9 us of a 64-frame block), so the steps between the array's fill and its drain -- n_ops - 1 ... frames - 1, when
			// every lane has a frame -- are kept short: the input arrives by ONE dpp move per half whose `old` operand is the frame from LDS (lane 0 of a
			// row has no lane below: it keeps `old`), sections update their states under the execution mask (a branch the compiler may not turn into four
			// selects: the empty asm), every lane stores (no branch), nobody asks who is active, and the frames come from LDS four steps ahead.
			auto step_any = [&](int t, double xin) {                 // fill and drain: lane j has a frame while 0 <= t - j < frames
				const double x = row_shr1_or(prev, xin);
				const double r = fma(a, x, s0);
				const bool active = (unsigned) (t - j) < (unsigned) nf;
				// biquad.h:76-92: r = c0 s + m0;  m0 = m1 + c1 s - c3 r;  m1 = c2 s - c4 r   (gain / add / pass: r = fma(a, x, b), no state)
				if (active && upd) { const double tt = fma(c1, x, m1), u = c2 * x; s0 = fma(-c3, r, tt); m1 = fma(-c4, r, u); }
				if (active && wr) wr_base[t * wr_stride] = r;
				prev = r;
			};
			// (fill and drain too ask for their frame a step ahead: a frame asked for where it is used is an LDS round trip per step -- 95 ns a step against 47)
			const int t_fill = (n_ops - 1 < steps) ? n_ops - 1 : steps;
			int t = 0;
			double xn = frame_in(0);
			for (; t < t_fill; ++t) { const double xc = xn; xn = frame_in(t + 1 < nf ? t + 1 : nf - 1); step_any(t, xc); }
			if (t < nf) {
				double xq[4];
#pragma unroll
				for (int q = 0; q < 4; ++q) xq[q] = frame_in(t + q < nf ? t + q : nf - 1);
				auto step_full = [&](int tt_, double xin) {
					const double x = row_shr1_or(prev, xin);
					const double r = fma(a, x, s0);
					if (upd) { asm volatile(""); const double tt = fma(c1, x, m1), u = c2 * x; s0 = fma(-c3, r, tt); m1 = fma(-c4, r, u); }
					wr_base[tt_ * wr_stride] = r;
					prev = r;
				};
				// (the frame four steps ahead by a running pointer, not clamped to the block: what it reads behind the last frame -- LDS, at worst beyond the
				// allocation, where a read gives zeros -- belongs to steps this loop does not run)
				const double *ahead = rd + (size_t) (t + 4) * C;
				for (; t + 4 <= nf; t += 4) {
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						step_full(t + q, xq[q]);
						xq[q] = ahead[q * C];
					}
					ahead += 4 * C;
				}
				xn = frame_in(t < nf ? t : nf - 1);
				for (; t < nf; ++t) { const double xc = xn; xn = frame_in(t + 1 < nf ? t + 1 : nf - 1); step_any(t, xc); }
			}
			// (behind the last frame lane 0 has nothing to take in: whatever it is handed is the input of steps nobody is active in)
			for (; t < steps; ++t) step_any(t, xn);
			if (upd) { st_agent(o.stp, s0); st_agent(o.stp + 1, m1); }
			}
		};
		if constexpr (!GEN) { cascade(op0, p.cs[0].C, p.cs[0].n_ops, m0a, m1a); __syncthreads(); }
		for (int k = 0; GEN && k < p.n_pass; ++k) {
			const ResidentPass &ps = p.pass[k];
			if (ps.kind == RES_PASS_REMIX) {
				// every output channel the sum of its input channels, in ascending order from 0.0, one rounding per sum (remix.c:39-101) -- or, weighted
				// (st2ms.c:34-38, crossfeed.c:41-46), the first product starts the sum and every operation rounds once: bit-exact either way
#pragma clang fp contract(off)
				const int ci = ps.c_in, co = ps.c_out, mn = ps.max_n;
				const double *tw = tab_lds + k * RES_TAB_DOUBLES, *tpost = tw + co * mn;
				const int *tidx = reinterpret_cast<const int *>(tw + co * mn + co);
				const bool weighted = ps.w != nullptr, has_post = ps.post != nullptr;
				for (int e = tid; e < frames * co; e += nth) {
					const int t = e / co, c = e - t * co;
					const int *idx = tidx + c * mn;
					double acc = 0.0;
					if (weighted) {
						const double *w = tw + c * mn;
						if (idx[0] >= 0) acc = cur[t * ci + idx[0]] * w[0];
						for (int q = 1; q < mn; ++q) { const int sc = idx[q]; if (sc < 0) break; const double prod = cur[t * ci + sc] * w[q]; acc = acc + prod; }
						if (has_post) acc = acc * tpost[c];
					}
					else for (int q = 0; q < mn; ++q) { const int sc = idx[q]; if (sc < 0) break; acc = acc + cur[t * ci + sc]; }
					oth[e] = acc;
				}
				double *sw = cur; cur = oth; oth = sw;
				__syncthreads();
			}
			else if (ps.kind == RES_PASS_FIR) {
				// direct form in the reference's order (fir.c:43-62): ((0 + x[t-T+1] h[T-1]) + ...) + x[t] h[0], products and sums rounded one by one
#pragma clang fp contract(off)
				const int cc = ps.c_in, T = ps.T;
				const double *hl = hist_lds + k * RES_FIR_MAX_CH * RES_FIR_TAPS;
				const double *taps = tab_lds + k * RES_TAB_DOUBLES;
				const int *foc = reinterpret_cast<const int *>(taps + RES_FIR_MAX_CH * RES_FIR_TAPS);
				for (int e = tid; e < frames * cc; e += nth) {
					const int t = e / cc, c = e - t * cc;
					const int fc = foc[c];
					double acc = cur[e];
					if (fc >= 0) {
						const double *h = taps + fc * RES_FIR_TAPS;
						acc = 0.0;
						if (t >= T - 1) {
							// (every tap's frame lies in the block: one pointer walking forward, no question asked per tap)
							const double *xp = cur + (t - (T - 1)) * cc + c;
#pragma unroll 4
							for (int m = T - 1; m >= 0; --m) { const double prod = xp[0] * h[m]; acc = acc + prod; xp += cc; }
						}
						else {
							for (int m = T - 1; m >= 0; --m) {
								const int ti = t - m;
								const double x = (ti >= 0) ? cur[ti * cc + c] : hl[c * RES_FIR_TAPS + (-ti - 1)];
								const double prod = x * h[m];
								acc = acc + prod;
							}
						}
					}
					oth[e] = acc;
				}
				// the history the next block starts from: slot q = x[-(q + 1)] of its first frame, i.e. the last T - 1 inputs of [old history | this block]
				double *hd = ps.hist + (size_t) ((low >> (16 + k)) & 1u) * cc * RES_FIR_TAPS;
				for (int e = tid; e < cc * (T - 1); e += nth) {
					const int c = e / (T - 1), q = e - c * (T - 1);
					if (foc[c] < 0) continue;
					st_agent(hd + c * RES_FIR_TAPS + q, (q < frames) ? cur[(frames - 1 - q) * cc + c] : hl[c * RES_FIR_TAPS + (q - frames)]);
				}
				double *sw = cur; cur = oth; oth = sw;
				__syncthreads();
			}
			else if (ps.kind == RES_PASS_CASCADE) {
				if (ps.casc == 0) cascade(op0, p.cs[0].C, p.cs[0].n_ops, m0a, m1a); else cascade(op1, p.cs[1].C, p.cs[1].n_ops, m0b, m1b);
				__syncthreads();
			}
		}
#ifdef RES_TIMING
		const unsigned long long tt2 = wall_clock64();
#endif
		for (int e = tid; e < n_out; e += nth) st_unit(r_out, e, cur[e], rq);
#ifdef RES_TIMING
		// (an experimental build, scripts/r06_resident_timing.sh: where a block's time goes, in ticks of the 100 MHz clock, summed in the control block)
		if (tid == 0) { const unsigned long long tt3 = wall_clock64(); p.ctl->pad[0] += (unsigned) (tt1 - tt0); p.ctl->pad[1] += (unsigned) (tt2 - tt1); p.ctl->pad[2] += (unsigned) (tt3 - tt2); p.ctl->pad[3] += 1; p.ctl->pad[4] += it - it_last; }
		it_last = it;
#endif
		__syncthreads();                                             // (the block buffer is free for the next block)
		done = seq;
		t_last = wall_clock64();
	}
	__threadfence_system();                                          // states (and whatever output is still on its way) are out before the word that says so
	__syncthreads();
	if (tid == 0) __hip_atomic_store(&p.ctl->alive, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

bool launch_cascade_resident(const ResidentParams &p, size_t lds_bytes, hipStream_t st)
{
	if (p.Cin < 1 || p.Cout < 1 || p.n_pass < 1 || p.n_pass > RES_MAX_PASSES || p.n_casc < 0 || p.n_casc > RES_MAX_CASCADES
	    || resident_lds_bytes(p.buf_doubles) > lds_bytes || (p.buf_doubles & 3) || !p.mail_in || !p.mail_out) return false;
	for (int q = 0; q < p.n_casc; ++q) if (p.cs[q].C < 1 || p.cs[q].C > 32 || p.cs[q].n_ops < 1 || p.cs[q].n_ops > RES_MAX_OPS) return false;

	// a wave per four channels of a cascade pass (a 16-lane row per channel); the other passes share the block out over whatever threads there are
	int widest = 1;
	for (int q = 0; q < p.n_casc; ++q) widest = p.cs[q].C > widest ? p.cs[q].C : widest;
	bool other = false;
	for (int k = 0; k < p.n_pass; ++k) {
		const ResidentPass &ps = p.pass[k];
		if (ps.kind == RES_PASS_REMIX && (size_t) ps.c_out * ps.max_n * 12 + (size_t) ps.c_out * 8 > (size_t) RES_TAB_DOUBLES * 8) return false;
		if (ps.kind == RES_PASS_FIR && (ps.c_in > RES_FIR_MAX_CH || ps.T > RES_FIR_TAPS)) return false;
		other |= ps.kind != RES_PASS_CASCADE;
	}
	// (passes that are not a systolic array share the block out over the threads there are: four waves for them)
	int waves = (widest + 3) / 4;
	if (other && waves < 4) waves = 4;
	if (!other && p.n_pass == 1) {
		grant_dynamic_lds(reinterpret_cast<const void *>(cascade_resident<false>), lds_bytes);
		hipLaunchKernelGGL(cascade_resident<false>, dim3(1), dim3(64 * waves), lds_bytes, st, p);
	}
	else {
		grant_dynamic_lds(reinterpret_cast<const void *>(cascade_resident<true>), lds_bytes);
		hipLaunchKernelGGL(cascade_resident<true>, dim3(1), dim3(64 * waves), lds_bytes, st, p);
	}
	return hipGetLastError() == hipSuccess;
}

}  // namespace dspamd
