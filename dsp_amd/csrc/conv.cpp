// conv.cpp -- FIR stages: FFT overlap-save convolver (fir / fir_p / hilbert / zita-equivalent) and the
// direct form for <= 32 taps.  Kernels: kernels_fft.hip.  Resampler: resample.cpp.
#include "stages.h"
#include "fft_params.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>

namespace dspamd {

Stage *make_resample_stage(const Spec &sp, int n_streams, ssize_t max_frames);

static long next_pow2(long v)
{
	long p = 1;
	while (p < v) p <<= 1;
	return p;
}

static int ilog2(long v)
{
	int l = 0;
	while ((1L << l) < v) ++l;
	return l;
}

static void make_twiddles(long n, long count, long stride, std::vector<double2> &out)
{
	// out[k] = exp(-2 pi i k stride / n), k < count
	out.resize(count);
	for (long k = 0; k < count; ++k) {
		const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double) ((k * stride) % n) / (long double) n;
		out[k] = make_double2((double) cosl(a), (double) sinl(a));
	}
}

// ------------------------------------------------------------------ ConvStage

class ConvStage : public Stage {
public:
	// ring_parent (tail child of a small-call stage): work on that stage's rings instead of own ones; force_N: transform size;
	// upc_block > 0: the uniformly partitioned form -- blocks of upc_block frames, transforms of 2 upc_block points, the filter in
	// partitions of upc_block taps with a frequency-domain delay line in the row kernel (conv_row mode 3)
	bool init(const Spec &sp, ssize_t max_frames, CascadeStage *feeder, Stage *prev, ConvStage *ring_parent = nullptr, long force_N = 0, long upc_block = 0);
	const char *type() const override { return "conv"; }
	std::string describe() const override;
	ssize_t run(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st) override;
	ssize_t max_out_frames(ssize_t in_frames) const override { return ((long long) in_frames * up + down - 1) / down; }
	ssize_t drain2(ssize_t max_frames, double *out, long out_stride, hipStream_t st) override;
	void reset(hipStream_t st) override;
	// K3 simply starts writing d frames later (k_origin): the 64 frames of a zita_convolver's latency used to cost a whole pass
	// over the output (9.8 of 43.5 ms on BASELINE config 5)
	bool absorb_discard(long d) override
	{
		// (a channel that is not convolved passes through the de-interleaving pass, which knows nothing of frames to drop)
		// (nor for a stage that reads the slab itself: K1 files only what lies behind first_n there -- dropped frames of a short call would never
		// reach the rings that later windows look back at.  A stage whose rings are written by the cascade or convolver in front never reads
		// the slab, whatever `direct` says: the same predicate as run()'s use_direct)
		if (resampler || feeds || fdl || !all_selected || (direct && !fed) || d <= 0) return false;
		skip = skip_left = d;
		return true;
	}
	// first stage of a pipeline fed in a wire format: the de-interleaving pass converts any format, K1 in direct mode the ones
	// whose channel pairs are naturally aligned
	bool wire_in_ok(int fmt, const void *in, long in_stride, ssize_t frames, bool also_out, int out_fmt) const override
	{
		(void) in; (void) in_stride; (void) frames; (void) out_fmt; (void) also_out;   // (K1 and K3 are different kernels: the ends are independent)
		if (!wire_fusion_on() || fed) return false;
		return !direct || (pcm_fusable(fmt) && fmt != PCM_DOUBLE);
	}
	// a plain convolution of every channel at the end of a pipeline: K3 applies the sink (dither, clip, wire format) in its stores
	bool wire_out_ok(int fmt, const void *out, long out_stride, ssize_t frames, bool also_in, int in_fmt) const override
	{
		(void) out_stride; (void) in_fmt; (void) also_in;
		if (!wire_fusion_on() || !pcm_fusable(fmt) || !all_selected || feeds) return false;
		// K3 speaks the formats in its plain form and in its two-phase form (the 2x upsampler, at least 3 pairs per stream)
		const bool plain = !resampler && nph == 1 && up == 1 && down == 1;
		// (the two-phase form writes whole pairs only: adjacent channels of an aligned slab)
		const bool twice = resampler && nph == 2 && up == 2 && down == 1 && pps >= 3 && !round_f32 && n_filters == 1 && (ch_in % 2) == 0 && ((((size_t) out) & 15) == 0);
		// any other phase count / ratio runs the general K3, which applies the sink sample by sample (what launch_col_inv_pps picks the
		// two-phase form for must match `twice`, whose sink writes whole pairs)
		const bool two_phase_form = resampler && nph == 2 && up == 2 && down == 1 && pps >= 3 && !round_f32;
		const bool general = resampler && !two_phase_form;
		if (!plain && !twice && !general) return false;
		// (the small-call regime writes through conv_fdl, which applies the sink like K3 -- pairs of adjacent channels or single ones)
		return true;
	}
	size_t device_bytes() const override
	{
		return (is_tail_child ? 0 : ring.bytes) + W.bytes + H.bytes + H_plain.bytes + tail_z.bytes + tail_scratch.bytes + tail_out.bytes
		       + fdl_buf.bytes + fdl_H.bytes + tail_buf.bytes + upc_buf.bytes + (tail_conv ? tail_conv->device_bytes() : 0)
		       + (upc_conv ? upc_conv->device_bytes() : 0);
	}
private:
	bool prepare_filters(const Spec &sp);
	void push(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st);
	void convolve(long q_lo, long q_hi, long k_origin, long out_count, double *out, long out_stride, hipStream_t st);
	ssize_t emit(long count, double *out, long out_stride, hipStream_t st);
	// integer-ratio resampling = nph = up filters (the polyphase branches) on the same input, outputs interleaved in
	// time (up > 1), or one filter whose output is kept every down-th frame (down > 1); see resample.cpp for the maths
	int up = 1, down = 1, nph = 1;
	bool resampler = false;
	long out_delay = 0, q_total = 0, emitted = 0, q_abs = 0;   // q_abs: absolute input index of the next frame (plain conv)
	std::vector<double> rs_tab;
	// merged fir_p + resample: the reference's fir_p stops producing when its input stops, so the resampler's drain2
	// tail sees ZEROS after the last fir_p frame, not the FIR tail the merged filter would add.  The tail is therefore
	// computed the reference's way: the last J fir_p outputs (plain filter h, H_plain) pushed through a polyphase
	// resampler of their own, whose drain2 is the tail.
	bool merged_pre = false;
	int J_rs = 0;
	DevBuf H_plain, tail_z, tail_scratch, tail_out;
	std::unique_ptr<Stage> tail_rs;
	long tail_frames = -1, tail_served = 0;
	std::vector<double> pre_taps;
	bool spectrum_of(const std::vector<double> &taps_1ch, long n_taps, int stride, int offset, double2 *dst, int row_nph);
	bool compute_tail(hipStream_t st);
	ConvParams base_params() const;
	long T = 0, N = 0, N1 = 0, N2 = 0, B = 0, first_n = 0, lat = 0, ring_len = 0, pos = 0, w_stride = 0, ring_stride = 0;   // ring_stride: distance between the rings of two pairs (ring_len + padding, see w_stride)
	int log2N1 = 0, log2N2 = 0, log2_lo = 0, nsel = 0, pps = 0, n_filters = 1, round_f32 = 0;
	bool fed = false, all_selected = false;
	// plain zero-latency convolution of all channels straight from the interleaved slab: K1 reads the new frames there and
	// files what later windows need in the ring itself -- no de-interleaving pass (6.4 GB of traffic at the headline shape)
	bool direct = false;
	const double *cur_slab = nullptr;
	long cur_slab_stride = 0, cur_q0 = 0;
	long pairs_per_chunk = 0;            // pairs whose W rows exist: every pair of the batch (launching a few streams at a time to keep W in the Infinity
	                                     // Cache, with or without side streams, never paid -- docs/history.md section 4.2 -- and is gone)
	std::string name;
	CascadeStage *feeder_ = nullptr;
	// this stage's K3 writes the next convolver's ring (set by the consumer's init)
	double2 *feed_ring = nullptr;
	long feed_stride = 0, feed_mask = 0, feed_pos = 0;
	int feed_round = 0;
	ConvStage *feeds = nullptr, *fed_by = nullptr;
	DevBuf ring, W, H, tw_n1, tw_n2, tw_hi, tw_lo, tw_col, pair_h, pair_out_ch, slot_of_channel;
	// float32 working precision (W, H, twiddle tables as float2; kernels_fft32.hip) -- the arithmetic of the reference's own
	// zita_convolver path (float32 in, float32 transforms, float32 out: zita_convolver.cpp:44,53,110).  Filter spectra are
	// still computed by the fp64 kernels (fp64 tables kept for that) and rounded once.
	bool f32 = false;
	DevBuf tw_n1f, tw_n2f, tw_colf;
	long skip = 0, skip_left = 0;      // leading output frames of the stream that are dropped (absorb_discard)
	// uniformly partitioned form (round 3): upc_P partitions of upc_B taps, spectra H[q][N], delay lines upc_buf[upc_P][S pps][N];
	// every block of upc_B frames must come through convolve() exactly once and in order (upc_slot = the slot the next one writes)
	int upc_P = 0, upc_slot = 0;
	long upc_B = 0, T_taps = 0;
	DevBuf upc_buf;
	size_t elem() const { return f32 ? sizeof(float2) : sizeof(double2); }
	bool spectrum_f32(const std::vector<double> &taps_1ch, long n_taps, int stride, int offset, size_t index, int row_nph);
	double2 *ring_dev = nullptr;         // ring.p, or the parent's rings (tail child)
	// ---- small-call regime (calls much shorter than the filter; the reference's own block is 2048 frames, dsp.h:38) ----
	// head: the first fD = fP1 x fB taps as a uniformly partitioned convolution with a frequency-domain delay line
	// (kernels_fft.hip conv_fdl; partition = fB frames, the largest power of two <= 2048 that divides the call size);
	// tail: the taps from fD on through a child overlap-save convolver on the SAME rings, run once per fD frames: at a
	// boundary n0 it computes (h_tail * x)[n0 - fD .. n0), which is the tail's share of the outputs n0 .. n0 + fD (tail_buf).
	// The rings of samples stay the state of truth: a call that does not fit the grid (other size, unaligned position)
	// simply continues on the one-transform-per-call path from the same rings, for the rest of the stream.
	bool fdl = false, fdl_live = false, is_tail_child = false;
	long fB = 0, fD = 0, fNF = 0;
	int fP1 = 0, f_slot = 0;
	DevBuf fdl_buf, fdl_H, fdl_tw, tail_buf;
	std::unique_ptr<ConvStage> tail_conv;
	bool init_fdl(const Spec &sp, ssize_t max_frames);
	// ---- mid-size calls (a multiple of a power of two F of at least 4096 frames, at most half the filter): the WHOLE filter in the
	// uniformly partitioned form with blocks of F frames -- a child stage on the same rings, transforms of 2 F points with one delay-line
	// slot per F taps, instead of one transform sized for the filter per call (whose hop the call fills to a fraction only).
	// As in the small-call regime the rings stay the state of truth: a call off the grid carries on with one transform per call.
	std::unique_ptr<ConvStage> upc_conv;
	bool upc_live = false;
	bool init_upc(const Spec &sp, ssize_t max_frames);
	void run_fdl(ssize_t frames, double *out, long out_stride, hipStream_t st);
	// ---- the cascade in front fused into this stage's first pass (kernels_fused.hip) ----
	// A call of exactly one hop on a window whose history is whole rows: the feeding cascade is not launched; fused_prepass +
	// cascade_chunk_carry find the section states at every row start, fused_col_fwd runs the sections from them in front of
	// its column transforms and files the last first_n frames of cascade output in the rings.  Everything else about the call
	// (K2, K3, the rings as state of truth, the cascade's carried state) is as on the separate path, which any other call takes.
	bool fuse_static = false;            // the plan allows it (decided once, in init)
	int fuse_seg = 1;
	bool fuse_accepts(const void *in, long in_stride, ssize_t frames, int in_fmt) const;
	bool fused_first_pass(const ConvParams &p, ssize_t frames, hipStream_t st);
	bool run_fused(ssize_t frames, double *out, long out_stride, hipStream_t st);
	bool fuse_this_call = false, fuse_failed = false;      // a resampler's call that convolve() starts with the fused first pass / that it could not
	// ... and with NO cascade in front (fir_p first in the chain, 8 channels, calls of one whole hop): the same first pass with one pass-through section
	// and all-zero states -- no prepass, no scan -- in place of K1's slab-direct form: two pairs of a frame per lane pair instead of one (K1 reads 16 of a
	// frame's 64 bytes per workgroup: 8.9 ms at the headline shape against 6.9)
	bool fuse_plain = false;
	DevBuf plain_sec, plain_op, plain_X;
	// ---- short filters behind long calls (round 5, kernels_short.hip): taps - 1 <= 4096 and calls of at least 1024 frames (<= 8192 where the calls fill the
	// larger window's blocks) -- the whole transform of a pair (8192 or 16384 points) in one workgroup's LDS: a block is one read of the window and one write of
	// the outputs instead of three trips of W through HBM
	bool short_mode = false;
	long short_N = CONV_SHORT_N;
	long cur_frames = 0;
	DevBuf tw_short;
	bool prepare_short(const Spec &sp);
	void convolve_short(long q_lo, long q_hi, long k_origin, long out_count, double *out, long out_stride, hipStream_t st);
	bool run_fused_plain(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st);
};

std::string ConvStage::describe() const
{
	std::ostringstream o;
	o << (resampler ? "fft-resample[" : "conv[") << name;
	if (resampler) o << " " << fs_in << "->" << fs_out << " " << up << "/" << down << " delay=" << out_delay;
	o << " T=" << T << " N=" << N << "=" << N1 << "x" << N2 << " hop=" << B << " pairs/stream=" << pps
	  << (n_filters > 1 ? " per-channel-filters" : "") << (lat ? " latency=" + std::to_string(lat) : "") << (fed ? (fed_by ? " fed-by-conv" : " fed-by-cascade") : "")
	  << (round_f32 ? " f32-io" : "") << (f32 ? " f32-spectrum" : "") << ((direct && !fed) ? (fuse_plain ? " slab-direct(two pairs per workgroup at whole hops)" : " slab-direct") : "");
	if (short_mode) o << " one-trip";
	if (fuse_static) o << " cascade-fused(" << (N1 - first_n / N2) * fuse_seg << " chunks of " << N2 / fuse_seg << (feeder_ && feeder_->fuse_tables().pairs > 1 ? ", sections per pair" : "") << ")";
	if (skip) o << " drops-first=" << skip;
	if (upc_conv) o << " mid-size-calls: " << upc_conv->upc_P << "x" << upc_conv->upc_B << " taps delay line N=" << upc_conv->N << (upc_conv->short_mode ? " one-trip" : "");
	if (fdl) {
		o << " small-calls: head " << fP1 << "x" << fB << " taps delay line";
		if (tail_conv && tail_conv->upc_P) o << " + tail " << tail_conv->upc_P << "x" << tail_conv->upc_B << " taps delay line N=" << tail_conv->N << (tail_conv->short_mode ? " one-trip" : "") << " per " << fD << " frames";
		else if (tail_conv) o << " + tail T=" << tail_conv->T << " N=" << tail_conv->N << " per " << fD << " frames";
	}
	o << "]";
	return o.str();
}

ConvParams ConvStage::base_params() const
{
	ConvParams p;
	memset(&p, 0, sizeof(p));
	p.log2N1 = log2N1; p.log2N2 = log2N2; p.log2_lo = log2_lo;
	p.N = N; p.N1 = N1; p.N2 = N2;
	p.ring = ring_dev;
	p.ring_row_stride = ring_stride; p.ring_mask = ring_len - 1;
	p.pair_h = pair_h.as<int>();
	p.shared_h = (n_filters == 1) ? 1 : 0;
	p.W = W.as<double2>();
	p.f32 = f32 ? 1 : 0;
	p.tw_n1 = (f32 ? tw_n1f : tw_n1).as<double2>(); p.tw_n2 = (f32 ? tw_n2f : tw_n2).as<double2>();
	p.tw_hi = tw_hi.as<double2>(); p.tw_lo = tw_lo.as<double2>(); p.tw_col = (f32 ? tw_colf : tw_col).as<double2>();
	p.H = H.as<double2>();
	p.h_scale = 1.0 / (double) N;
	p.C = ch_in;
	p.pairs_per_stream = pps;
	p.pair_out_ch = pair_out_ch.as<int>();
	p.round_f32 = round_f32;
	p.sink = { 0, PCM_DOUBLE, 0.0, 0, nullptr };
	p.slab_fmt = PCM_DOUBLE;
	p.nph = nph; p.up = up; p.down = down;
	p.w_stride = w_stride;
	p.phase_stride = pairs_per_chunk * w_stride;
	p.ring_out = feed_ring;
	p.ring_out_stride = feed_stride; p.ring_out_mask = feed_mask; p.ring_out_pos = feed_pos;
	p.ring_out_round_f32 = feed_round;
	// non-temporal hints: K1's ring loads and W stores (data touched once per launch): 5.95 -> 5.55 ms at the headline shape; on
	// K2 and K3 they measured nothing (scripts/exp_nt.sh)
	p.nt = 3;      // (non-temporal K1 ring loads and W stores: docs/history.md section 4.2)
	return p;
}

// Transform size for a T-tap filter and calls of max_frames frames: at least 2T (overlap <= 1/2), up to 16x the
// filter when calls are long (valid fraction (N - T + 1) / N: 1/2 at 2T, 15/16 at 16T); among the admissible sizes
// the cheapest for such a call: blocks x points x relative cost per point of the three kernels together (measured with the
// persistent row kernel, round 2: K1 + K2 + K3 = 1.55 + 1.83 + 1.47 per 2^28 points at 1024-point rows; K2 2.2 at 2048- and
// 4096-point rows -> 1.08).  *cost (optional) = that figure, comparable between plans.
long conv_plan(long T, long max_frames, bool resampler, double *cost_out)
{
	const long lo = std::max<long>(next_pow2(2 * T), 1L << (FFT_MIN_LOG2_N2 + FFT_MIN_LOG2_N1));
	const long fn = (T - 1 + 7) & ~7L, F = std::max<long>(max_frames, 1);
	const long want = next_pow2(fn + F);
	const long hi = std::min(std::min(std::max(lo, want), std::max(lo * 8, resampler ? (1L << 16) : 0L)), 1L << (FFT_MAX_LOG2_N1 + FFT_MAX_LOG2_N2));
	long N = lo;
	double best = 0.0;
	for (long n = lo; n <= std::max(lo, hi); n <<= 1) {
		const long hop = (n - fn) & ~7L;
		if (hop <= 0) continue;
		const long n2 = n / std::min<long>(1L << FFT_MAX_LOG2_N1, ((n >> 10) >= (1L << FFT_MIN_LOG2_N1)) ? (n >> 10) : (n >> FFT_MIN_LOG2_N2));
		const double c = (n2 >= 2048) ? 1.08 : 1.0;
		const double cost = (double) ((F + hop - 1) / hop) * ((double) n * c + 16384.0);   // + per-block launch / tail overhead
		if (best == 0.0 || cost < best) { best = cost; N = n; }
	}
	if (cost_out) *cost_out = best;
	return N;
}

bool ConvStage::init(const Spec &sp, ssize_t max_frames, CascadeStage *feeder, Stage *prev, ConvStage *ring_parent, long force_N, long upc_block)
{
	is_tail_child = ring_parent != nullptr;
	name = sp.name;
	T = T_taps = sp.T;
	if (upc_block > 0) {
		// the window of a block is [the block before | the block]: as an overlap-save plan that is a filter of upc_block + 1 taps
		// (first valid output = upc_block, hop = upc_block) on a transform of 2 upc_block points
		upc_B = upc_block;
		upc_P = (int) ((sp.T + upc_B - 1) / upc_B);
		T = upc_B + 1;
		force_N = 2 * upc_B;
	}
	lat = sp.latency;
	round_f32 = (sp.conv_mode == CONV_ZITA_EQUIV);
	f32 = round_f32 && !getenv("DSP_AMD_ZITA_F64");      // (the switch keeps fp64 transforms behind the float32 I/O: round 2's form)
	nsel = num_set(sp.sel);
	all_selected = (nsel == ch_in);
	n_filters = (sp.fch == 1) ? 1 : nsel;
	if (sp.kind == Kind::Resample) {
		// resample acts on every channel whatever the selector says (README.md:389-391)
		resampler = true;
		up = sp.rs_n; down = sp.rs_d; nph = up;
		int J = 0;
		resample_polyphase_table(sp, &J, &out_delay, rs_tab);
		T = J;
		if (!sp.rs_pre.empty()) {
			// branch p of the merged stage = pre * h_p
			const long Tp = (long) sp.rs_pre.size(), Tm = Tp + J - 1;
			std::vector<double> g((size_t) Tm * up, 0.0);
			for (int ph = 0; ph < up; ++ph)
				for (long i = 0; i < Tp; ++i) {
					const double a = sp.rs_pre[i];
					if (a == 0.0) continue;
					for (int j = 0; j < J; ++j) g[(size_t) (i + j) * up + ph] += a * rs_tab[(size_t) j * up + ph];
				}
			rs_tab.swap(g);
			T = Tm;
			name = sp.rs_pre_name + "+" + sp.name;
			merged_pre = true;
			J_rs = J;
			pre_taps = sp.rs_pre;
			Spec plain(sp);
			plain.rs_pre.clear();
			tail_rs.reset(make_resample_stage(plain, S, J + 8));
			if (!tail_rs) return false;
			tail_rs->S = S; tail_rs->ch_in = sp.ch_in; tail_rs->ch_out = sp.ch_out; tail_rs->fs_in = sp.fs_in; tail_rs->fs_out = sp.fs_out;
		}
		lat = 0;
		nsel = ch_in; all_selected = true; n_filters = 1;
	}
	// channel pairs share a transform only when they share the filter
	pps = (n_filters == 1) ? (nsel + 1) / 2 : nsel;

	const char *env = getenv("DSP_AMD_CONV_LOG2N");
	{
		const char *se = getenv("DSP_AMD_CONV_SHORT");          // 0 = the four-step transforms whatever the filter's length; 13 / 14 = the 8192- / 16384-point window where either would do (read per stage: the tests build the plans in one process)
		const int sv = se ? atoi(se) : 1;
		const long fn = (T - 1 + 7) & ~7L;
		// (calls of at least 1024 frames: below that a launch is all latency; 32-bit byte offsets inside a stream's slab / a pair's ring)
		const bool short_ok = sv != 0 && !env && !resampler && !round_f32 && !ring_parent && !upc_block && !force_N
		                      && (long) max_frames >= 1024 && (long) max_frames <= (1L << 24) && (double) max_frames * sp.ch_in * sizeof(double) < 2.0e9;
		// the window: a block of the 16384-point form is 14 / 13 of the transform work per point for (16384 - fn) instead of (8192 - fn) outputs, and
		// (N + hop) / hop units of traffic -- measured on BASELINE config 5's shape a block costs 51 ns of the chip on the small window (two workgroups per
		// CU) and 122 ns on the large one: the same per frame at first_n = 2308 --, provided the calls fill its blocks; filters of 4098 ... 8193 taps have
		// that window or the four-step transforms
		const bool fits13 = fn <= CONV_SHORT_N / 2, fits14 = fn <= CONV_SHORT_N2 / 2, fill14 = (long) max_frames >= 4 * (CONV_SHORT_N2 - fn);
		const bool want14 = fits14 && sv != 13 && (fits13 ? (sv == 14 || (fn > 2304 && fill14)) : fill14);
		short_mode = short_ok && (fits13 || want14);
		short_N = want14 ? CONV_SHORT_N2 : CONV_SHORT_N;
		// the uniformly partitioned form on windows of 8192 / 16384 points (the delay-line tail of the small-call regime at the headline: 7 slots of
		// 8192 taps; calls that are multiples of 4096 / 8192 frames): the same kernel with the delay line between its two transforms -- one launch per
		// call in place of K1 / K2 (mode 3) / K3 per block, the two trips of W gone (16-byte offsets inside a ring row: 32 bits)
		if (sv != 0 && !env && upc_block && (2 * upc_block == CONV_SHORT_N || 2 * upc_block == CONV_SHORT_N2) && !resampler && !round_f32 && !f32 && nph == 1) {
			short_mode = true;
			short_N = 2 * upc_block;
		}
	}
	N = short_mode ? short_N : conv_plan(T, max_frames, resampler, nullptr);
	const long lo = std::max<long>(next_pow2(2 * T), 1L << (FFT_MIN_LOG2_N2 + FFT_MIN_LOG2_N1));
	if (env && !is_tail_child) N = std::max(lo, 1L << atoi(env));
	if (force_N) N = force_N;
	if (N > (1L << (FFT_MAX_LOG2_N1 + FFT_MAX_LOG2_N2))) N = std::max(lo, 1L << (FFT_MAX_LOG2_N1 + FFT_MAX_LOG2_N2));
	// N = N1 x N2: columns (strided) 16..256 points (one LDS exchange), rows (contiguous) 512..4096 points
	// 1024-point rows where possible: one wave owns a row there and K2 needs no workgroup barrier (the best-tuned geometry)
	N1 = std::min<long>(1L << FFT_MAX_LOG2_N1, ((N >> 10) >= (1L << FFT_MIN_LOG2_N1)) ? (N >> 10) : (N >> FFT_MIN_LOG2_N2));
	N2 = N / N1;
	log2N1 = ilog2(N1);
	log2N2 = ilog2(N2);
	if (log2N2 > FFT_MAX_LOG2_N2) {
		set_error("%s: error: filter too long for the GPU convolver (%ld taps; limit %ld)", name.c_str(), T, (1L << (FFT_MAX_LOG2_N2 + FFT_MAX_LOG2_N1 - 1)));
		return false;
	}
	// The window starts up to 7 samples earlier than overlap-save needs, so that the first valid output index
	// (first_n) and the hop are multiples of 8: K3's runs of 8 frames then start on 512-byte boundaries of the slab.
	first_n = (T - 1 + 7) & ~7L;
	B = (N - first_n) & ~7L;
	// behind a cascade, with 256-row windows: when the calls are exactly what is left of the window after 16 (or 32) WHOLE rows of history, take
	// that much history -- more of it than overlap-save needs costs nothing (the hop is the call either way), and the cascade can then be fused
	// into the first pass (below): a 50000-tap filter on 983040-frame calls is served like the 65536-tap one
	// (the same for a convolver that reads its slab directly -- fir_p first in the chain: its whole-hop calls take the fused first pass too)
	// (round 5: any whole number of history rows up to 32 -- 17 rows behind the 66119 taps of fir_p merged into a 2x resampler, BASELINE config 4 at
	// 978944-frame calls: K3's output mapping counts from first_n whatever it is)
	const bool slab_direct_plan = !feeder && all_selected && n_filters == 1 && (ch_in % 2) == 0 && !round_f32 && sp.kind != Kind::Resample && !getenv("DSP_AMD_CONV_NO_DIRECT");
	if ((feeder || slab_direct_plan) && !ring_parent && !upc_block && !force_N && log2N1 == 8 && lat == 0 && (sp.kind != Kind::Resample || down == 1)) {
		const long rows = (N - (long) max_frames) / N2;
		if (rows >= 1 && rows <= 32 && rows * N2 >= first_n && N - rows * N2 == (long) max_frames) { first_n = rows * N2; B = N - first_n; }
	}
	ring_len = next_pow2(first_n + lat + std::max<long>(max_frames, B));
	log2_lo = (ilog2(N) + 1) / 2;

	// rings: one row of complex samples (x_a[n], x_b[n]) per channel pair per stream -- the sequence K1 transforms
	if (ring_parent) {
		if (ring_parent->ring_len < ring_len || ring_parent->pps != pps) { set_error("%s: BUG: tail convolver does not fit its parent's rings", name.c_str()); return false; }
		ring_len = ring_parent->ring_len;
		ring_stride = ring_parent->ring_stride;
		ring_dev = ring_parent->ring_dev;
	}
	else {
		ring_stride = ring_len + 272;       // (4352 bytes: pair rows off power-of-two distances, profiles/r02_tcc_padding.json)
		if (!ring.alloc((size_t) S * pps * ring_stride * sizeof(double2))) return false;
		ring_dev = ring.as<double2>();
	}
	std::vector<int> soc(ch_in, -1);
	std::vector<int> sel_ch;
	for (int c = 0; c < ch_in; ++c) if (resampler || sp.sel[c]) sel_ch.push_back(c);
	std::vector<int> ph((size_t) S * pps), poc((size_t) pps * 2);
	for (int q = 0; q < pps; ++q) {
		const int ia = (n_filters == 1) ? 2 * q : q, ib = (n_filters == 1 && 2 * q + 1 < nsel) ? 2 * q + 1 : -1;
		poc[2 * q] = sel_ch[ia];
		poc[2 * q + 1] = (ib >= 0) ? sel_ch[ib] : -1;
		soc[sel_ch[ia]] = 2 * q;
		if (ib >= 0) soc[sel_ch[ib]] = 2 * q + 1;
		for (int s = 0; s < S; ++s) ph[(size_t) s * pps + q] = (n_filters == 1) ? 0 : q;
	}
	direct = all_selected && n_filters == 1 && (ch_in % 2) == 0 && !round_f32 && lat == 0 && !getenv("DSP_AMD_CONV_NO_DIRECT");
	if (!slot_of_channel.upload(soc.data(), soc.size() * sizeof(int))) return false;
	if (!pair_h.upload(ph.data(), ph.size() * sizeof(int))) return false;
	if (!pair_out_ch.upload(poc.data(), poc.size() * sizeof(int))) return false;

	std::vector<double2> t;
	make_twiddles(N1, N1, 1, t);
	if (!tw_n1.upload(t.data(), t.size() * sizeof(double2))) return false;
	make_twiddles(N2, N2, 1, t);
	if (!tw_n2.upload(t.data(), t.size() * sizeof(double2))) return false;
	if (short_mode) { make_twiddles(N, N, 1, t); if (!tw_short.upload(t.data(), t.size() * sizeof(double2))) return false; }
	make_twiddles(N, 1L << log2_lo, 1, t);
	if (!tw_lo.upload(t.data(), t.size() * sizeof(double2))) return false;
	make_twiddles(N, N >> log2_lo, 1L << log2_lo, t);
	if (!tw_hi.upload(t.data(), t.size() * sizeof(double2))) return false;
	{
		// the inter-pass twiddle as the column kernels read it (kernels_fft.hip, col_twiddle): row j < P = N1 / 16 holds
		// w_N^(n2 j), row P holds w_N^(n2 P) -- two contiguous 16-byte look-ups per thread, the other rows by products
		const long P = N1 / 16;
		std::vector<double2> tc((size_t) (P + 1) * N2);
		for (long jj = 0; jj <= P; ++jj)
			for (long n2 = 0; n2 < N2; ++n2) {
				const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double) ((n2 * jj) % N) / (long double) N;
				tc[(size_t) jj * N2 + n2] = make_double2((double) cosl(a), (double) sinl(a));
			}
		if (!tw_col.upload(tc.data(), tc.size() * sizeof(double2))) return false;
		if (f32) {
			auto as_f32 = [](const std::vector<double2> &d) { std::vector<float2> f(d.size()); for (size_t i = 0; i < d.size(); ++i) f[i] = make_float2((float) d[i].x, (float) d[i].y); return f; };
			std::vector<double2> t1, t2;
			make_twiddles(N1, N1, 1, t1);
			make_twiddles(N2, N2, 1, t2);
			const std::vector<float2> f1 = as_f32(t1), f2 = as_f32(t2), fc = as_f32(tc);
			if (!tw_n1f.upload(f1.data(), f1.size() * sizeof(float2)) || !tw_n2f.upload(f2.data(), f2.size() * sizeof(float2)) ||
			    !tw_colf.upload(fc.data(), fc.size() * sizeof(float2))) return false;
		}
	}

	pairs_per_chunk = (long) S * pps;
	w_stride = N + 272;                     // (the same padding between the pairs of W)
	// (at least one fp64 row set: the filter spectra of a float32 stage are computed by the fp64 kernels in this buffer)
	// (the one-trip kernel keeps a pair's window in registers: no W)
	if (!W.alloc(std::max(short_mode ? (size_t) 0 : (size_t) nph * pairs_per_chunk * w_stride * elem(), (size_t) nph * w_stride * sizeof(double2)), false)) return false;
	if (!H.alloc((size_t) (upc_P ? (size_t) upc_P * n_filters : (size_t) n_filters * nph) * N * elem(), false)) return false;
	if (upc_P && !upc_buf.alloc((size_t) upc_P * S * pps * N * elem())) return false;      // (a float32 stage's delay line holds float2: half the traffic)
	log_msg(LL_VERBOSE, "%s: info: device buffers ring %p (%zu MB) W %p (%zu MB) H %p", name.c_str(), ring_dev, ring.bytes >> 20, W.p, W.bytes >> 20, H.p);
	if (!prepare_filters(sp)) return false;

	// a cascade directly in front may write the planar rings itself (saves one interleaved round trip)
	// another FFT convolver right before, on the same channel pairs: its K3 writes (re, im) = this ring's elements
	if (ConvStage *pc = ring_parent ? nullptr : dynamic_cast<ConvStage *>(prev)) {
		if (all_selected && n_filters == 1 && pc->all_selected && pc->n_filters == 1 && pc->ch_out == ch_in && pc->pps == pps && !pc->feeds
		    && !pc->merged_pre && !getenv("DSP_AMD_NO_FEED")) {
			pc->feed_ring = ring_dev;
			pc->feed_stride = ring_stride;
			pc->feed_mask = ring_len - 1;
			pc->feed_pos = 0;
			pc->feed_round = round_f32;
			pc->feeds = this;
			fed_by = pc;
			fed = true;
		}
	}
	if (feeder && all_selected && !round_f32 && feeder->Cg == ch_in && !getenv("DSP_AMD_NO_FEED")) {
		feeder->ring.base = reinterpret_cast<double *>(ring_dev);
		feeder->ring.row_stride = ring_stride;
		feeder->ring.mask = ring_len - 1;
		feeder->ring.pos = 0;
		feeder->ring.pair_ch = pair_out_ch.as<int>();
		feeder->ring.rows_per_stream = pps;
		feeder->ring.consecutive_pairs = (n_filters == 1 && (ch_in % 2) == 0) ? 1 : 0;
		feeder->write_interleaved = 0;
		feeder_ = feeder;
		fed = true;
	}
	if (!ring_parent && !init_upc(sp, max_frames)) return false;
	if (!ring_parent && !upc_conv && !init_fdl(sp, max_frames)) return false;
	// the cascade fused into the first pass: a uniform chain of sections in front, 256 rows whose first first_n / N2 are history -- at most
	// an eighth of the window: the fused pass works on the WINDOW (256 rows whatever the hop), the separate cascade on the hop, and with
	// 64 history rows (the headline chain at 196608-frame calls, N = 2^18) the fused kernels measured 3.41 ms against 3.09
	{
		const char *fe = getenv("DSP_AMD_FUSE");          // 0 = the separate kernels always (read per stage: the tests build both plans in one process)
		const bool fuse_on = !fe || atoi(fe) != 0;
		const long hist_rows = (N2 > 0 && first_n % N2 == 0) ? first_n / N2 : 0;
		if (fuse_on && feeder_ && !ring_parent)
			log_msg(LL_VERBOSE, "%s: info: fused first pass: regimes %d%d kind %d%d%d%d%d%d rows %d pairs %d%d hist %ld hop %d whole %d%d sizes %d%d sections %d", name.c_str(),
			        !upc_conv, !fdl, !resampler, nph == 1, n_filters == 1, !f32, !round_f32, lat == 0, log2N1 == 8, (pps % 2) == 0, ch_in == 2 * pps, hist_rows, B == N - first_n,
			        true, true, (double) B * ch_in * sizeof(double) < 2.0e9, (double) (2 * w_stride + N) * sizeof(double2) < 2.0e9, (int) feeder_->fuse_tables().ok);
		// (a resampler by an integer factor `up` -- nph = up branches on the same first pass, K2 / K3 in their multi-phase forms -- takes the same
		// first pass: only the calls whose outputs start at phase 0 of the call's first frame are whole windows, fuse_accepts looks at that)
		if (fuse_on && feeder_ && !ring_parent && !upc_conv && !fdl && (resampler ? (down == 1 && nph == up) : nph == 1) && n_filters == 1 && !f32 && !round_f32 && lat == 0
		    && log2N1 == 8 && (pps % 2) == 0 && ch_in == 2 * pps && hist_rows >= 1 && hist_rows <= 32 && B == N - first_n
		    && (double) B * ch_in * sizeof(double) < 2.0e9 && (double) (2 * w_stride + N) * sizeof(double2) < 2.0e9 && feeder_->fuse_tables().ok) {
			// row segments: enough workgroups for the chip when the streams are few, as far as the scan over the chunks fits its workgroup
			const int D = 2 * feeder_->n_ops;
			const long groups = (long) S * (pps / 2);
			fuse_seg = 1;
			auto scan_fits = [&](long k) { long g = 1; while (g * g < k) ++g; return (double) (k * D + ((k + g - 1) / g) * D + D * D) * sizeof(double) <= 160.0 * 1024 - 256 && ((k + g - 1) / g) * D <= 1024; };
			while (fuse_seg < 4 && groups * fuse_seg < 224 && (N2 / 8) % (4 * fuse_seg) == 0 && scan_fits((N1 - hist_rows) * 2 * fuse_seg)) fuse_seg *= 2;
			if (scan_fits((N1 - hist_rows) * fuse_seg)) {
				fuse_static = true;
				feeder_->fuse_probe = [this](const void *in, long in_stride, ssize_t frames, int in_fmt) { return fuse_accepts(in, in_stride, frames, in_fmt); };
			}
		}
		if (fuse_on && !feeder_ && !fed && !ring_parent && direct && !upc_conv && !fdl && !resampler && nph == 1 && n_filters == 1 && !f32 && !round_f32 && lat == 0
		    && log2N1 == 8 && (pps % 2) == 0 && ch_in == 2 * pps && hist_rows >= 1 && hist_rows <= 32 && B == N - first_n
		    && (double) B * ch_in * sizeof(double) < 2.0e9 && (double) (2 * w_stride + N) * sizeof(double2) < 2.0e9 && fused_section_slots(1) == 1) {
			const long groups = (long) S * (pps / 2);
			fuse_seg = 1;
			while (fuse_seg < 4 && groups * fuse_seg < 224 && (N2 / 8) % (4 * fuse_seg) == 0) fuse_seg *= 2;
			const double one[6] = { 1.0, 0.0, 0.0, 0.0, 0.0, 0.0 };        // r = s + m0;  m0 = m1;  m1 = 0: the sample itself, bit for bit, from zero states
			const int op0 = 0;
			const long K = (N1 - hist_rows) * fuse_seg;
			if (plain_sec.upload(one, sizeof(one)) && plain_op.upload(&op0, sizeof(op0)) && plain_X.alloc((size_t) S * K * ch_in * 2 * sizeof(double), true)) fuse_plain = true;
			else { (void) hipGetLastError(); plain_sec.release(); plain_op.release(); plain_X.release(); }
		}
	}
	return true;
}

bool ConvStage::fuse_accepts(const void *in, long in_stride, ssize_t frames, int in_fmt) const
{
	(void) in_stride;
	if (!(fuse_static && frames == B && skip_left == 0 && !feeds && ((resampler ? q_total : q_abs) & 7) == 0)) return false;
	// a resampler: the next output frame must be phase 0 of this call's first input frame (true from the second call on: the first one drops
	// out_delay outputs and its windows start inside the history) -- then this call is one whole window and emits up x frames outputs
	if (resampler && (emitted + out_delay) != q_total * up) return false;
	if (in_fmt == PCM_DOUBLE) { if ((((size_t) in) & 15) != 0) return false; }
	else {
		// a wire format (the cascade is the first stage of a pipeline run from format to format): read by the matrix-core prepass and by the first
		// pass themselves -- 8 channels, naturally aligned pairs
		if (!(wire_fusion_on() && pcm_fusable(in_fmt) && ch_in == 8 && feeder_->fuse_tables().n_real <= 16 && feeder_->fuse_tables().pairs == 1 && (((size_t) in) & 7) == 0)) return false;
	}
	// everything run_fused() will need exists before the answer is yes (the chunk plan's state buffers scale with S K C D; a plan is built
	// once per call shape and kept): after a yes the cascade launches nothing, so a failure in there could no longer be served by the
	// separate kernels -- a no here still can
	const long hist_rows = first_n / N2;
	CascadeStage::ChunkPlan *plan = feeder_->chunk_plan_for(frames, (int) ((N1 - hist_rows) * fuse_seg), N2 / fuse_seg);
	if (!plan) { (void) hipGetLastError(); return false; }
	if (in_fmt != PCM_DOUBLE && !feeder_->fuse_gtable(*plan)) { (void) hipGetLastError(); return false; }
	return true;
}

bool ConvStage::run_fused_plain(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st)
{
	const long hist_rows = first_n / N2;
	FuseParams f;
	memset(&f, 0, sizeof(f));
	f.in = in; f.in_stride_frames = in_stride; f.in_fmt = PCM_DOUBLE;
	f.C = ch_in; f.n_sec = 1; f.n_ops = 1;
	f.sec_op = plain_op.as<int>();
	f.gain = 1.0;
	f.seg = fuse_seg; f.hist_rows = (int) hist_rows;
	f.K = (N1 - hist_rows) * fuse_seg; f.len = N2 / fuse_seg;
	f.X = plain_X.as<double>();
	f.n_streams = S;
	ConvParams p = base_params();
	p.win_base = (q_abs - first_n) & (ring_len - 1);
	p.first_n = first_n;
	p.valid = N;
	p.out = out;
	p.out_stride_frames = out_stride;
	p.sink = wire_sink;
	p.in_count = frames;
	p.q_blk = q_abs;
	p.k_origin = q_abs;
	p.out_count = frames;
	p.k3_pipe_ok = (all_selected && n_filters == 1 && pps == 4 && ch_in == 8 && !feeds && ((((size_t) out) & 15) == 0)) ? 1 : 0;
	p.pair0 = 0; p.stream0 = 0; p.n_streams_launch = S;
	{ ProfScope ps("fused_col_fwd", st); if (!launch_fused_col_fwd(p, f, plain_sec.as<double>(), st)) return false; }
	{ ProfScope ps("conv_row", st); launch_conv_row(p, 0, (int) ((long) S * pps), st); }
	{ ProfScope ps("conv_col_inv", st); launch_conv_col(p, true, (int) ((long) S * pps), st); }
	return true;
}

// the first pass of a window whose new rows are the cascade's pending call (fuse_accepts said yes to exactly this call): the chunks' end states
// from zero state, the scan, fused_col_fwd -- in K1's place; p describes the window as for K1
bool ConvStage::fused_first_pass(const ConvParams &p, ssize_t frames, hipStream_t st)
{
	const CascadeStage::Pending pd = feeder_->pending;
	feeder_->pending = CascadeStage::Pending();
	const CascadeStage::FuseTables &ft = feeder_->fuse_tables();
	const long hist_rows = first_n / N2, len = N2 / fuse_seg, K = (N1 - hist_rows) * fuse_seg;
	CascadeStage::ChunkPlan *plan = feeder_->chunk_plan_for(frames, (int) K, len);
	if (!plan) return false;
	FuseParams f;
	memset(&f, 0, sizeof(f));
	f.in = pd.in; f.in_stride_frames = pd.in_stride; f.in_fmt = pd.in_fmt;
	f.C = ch_in; f.n_sec = ft.n_sec; f.n_ops = feeder_->n_ops;
	f.sec_op = ft.sec_op.as<int>();
	f.gain = ft.gain;
	if (ft.pairs > 1) { f.sec_stride = (long) ft.n_sec * 6; f.gain_tab = ft.gain_tab.as<double>(); }
	f.seg = fuse_seg; f.hist_rows = (int) hist_rows;
	f.K = K; f.len = len;
	f.cstate = plan->cstate.as<double>(); f.X = plan->X.as<double>();
	f.n_streams = S;
	{
		// the chunks' end states from zero state: as a product on the matrix cores where the shape allows (8 channels), else by the recurrence
		ProfScope ps("fused_prepass", st);
		if (!feeder_->fuse_gtable(*plan) || !launch_fused_prepass_mm(f, plan->G.as<double>(), N2, plan->g_states, st)) {
			if (f.in_fmt != PCM_DOUBLE || !launch_fused_prepass(f, ft.sec.as<double>(), N2, pps, st)) return false;       // (fuse_accepts lets a wire format through only where the matrix-core form serves it)
		}
		else ps.rename("fused_prepass_mm");
	}
	ChunkParams cp;
	memset(&cp, 0, sizeof(cp));
	cp.len = len; cp.C = ch_in; cp.K = (int) K; cp.D = 2 * feeder_->n_ops; cp.n_pow = plan->n_pow; cp.n_cls = plan->n_cls;
	cp.cls = plan->cls.as<int>(); cp.H = plan->H.as<double>(); cp.Mp = plan->Mp.as<double>();
	cp.cstate = plan->cstate.as<double>(); cp.X = plan->X.as<double>(); cp.state = feeder_->state.as<double>();
	{ ProfScope ps("cascade_chunk_carry", st); launch_chunk_carry(cp, S, st); }
	{ ProfScope ps("fused_col_fwd", st); if (!launch_fused_col_fwd(p, f, ft.sec.as<double>(), st)) return false; }
	if (plan->done) (void) hipEventRecord(plan->done, st);
	feeder_->ring.pos = (feeder_->ring.pos + frames) & feeder_->ring.mask;
	return true;
}

bool ConvStage::run_fused(ssize_t frames, double *out, long out_stride, hipStream_t st)
{
	ConvParams p = base_params();
	p.win_base = (q_abs - first_n) & (ring_len - 1);
	p.first_n = first_n;
	p.valid = N;
	p.out = out;
	p.out_stride_frames = out_stride;
	p.sink = wire_sink;
	p.in_count = frames;
	p.q_blk = q_abs;
	p.k_origin = q_abs;
	p.out_count = frames;
	p.k3_pipe_ok = (all_selected && n_filters == 1 && pps == 4 && ch_in == 8 && !feeds && ((((size_t) out) & 15) == 0)) ? 1 : 0;
	p.pair0 = 0; p.stream0 = 0; p.n_streams_launch = S;
	if (!fused_first_pass(p, frames, st)) return false;
	{ ProfScope ps("conv_row", st); launch_conv_row(p, 0, (int) ((long) S * pps), st); }
	{ ProfScope ps("conv_col_inv", st); launch_conv_col(p, true, (int) ((long) S * pps), st); }
	return true;
}

// ---- mid-size calls: see the member comment.  Traffic per pair and frame in units of 32 bytes: P + 5.5 (P delay-line slots read or
// written by the row kernel, 5.5 trips of the block's transform through HBM) against 3.25 N / frames for one transform of N
// points per call (less what K1 does not read and K3 does not write) and about 18.5 for the small-call regime's two levels (which
// wins from 16 slots on: measured at 4096-frame calls on 65536 taps, scripts/exp_mid.sh): chosen when 2 <= P = ceil(T / F) <= 12
// (16 where the calls are too long for the small-call regime) and it moves less than one transform per call.
bool ConvStage::init_upc(const Spec &sp, ssize_t max_frames)
{
	static const int max_slots = [] { const char *e = getenv("DSP_AMD_CONV_UPC"); return (!e || atoi(e) == 1) ? 12 : atoi(e); }();   // 0 = never, n > 1 = up to n slots
	if (max_slots < 2 || resampler || nph != 1 || merged_pre) return true;      // (`fir`'s / zita's latency is the child's too: its windows start lat frames earlier)
	// the block: the largest power of two that divides the call size (a call is then a whole number of blocks)
	long F = 1L << (FFT_MAX_LOG2_N2 + FFT_MAX_LOG2_N1 - 1);
	while (F > 1 && (max_frames % F)) F >>= 1;
	if (F < (1L << (FFT_MIN_LOG2_N2 + FFT_MIN_LOG2_N1 - 1)) || 2 * F > T) return true;
	const long slots = (T + F - 1) / F;
	// one transform per call: K1 reads the window's T + frames samples and writes N points, K2 reads and writes them, K3 reads them
	const double one = (4.0 * (double) N + (double) T + (double) max_frames) / (2.0 * (double) max_frames) + 0.5;
	const bool small_calls_possible = (long) max_frames * 8 <= T;
	if (slots > (small_calls_possible ? max_slots : std::max(max_slots, 16)) || (double) slots + 5.5 >= 0.95 * one) return true;
	upc_conv.reset(new ConvStage);
	upc_conv->S = S; upc_conv->ch_in = ch_in; upc_conv->ch_out = ch_out; upc_conv->fs_in = fs_in; upc_conv->fs_out = fs_out;
	Spec us(sp);
	us.name = sp.name + ":blocks";
	if (!upc_conv->init(us, F, nullptr, nullptr, this, 0, F)) {
		log_msg(LL_VERBOSE, "%s: info: mid-size-call regime not available (%s)", name.c_str(), last_error());
		upc_conv.reset();
		return true;
	}
	upc_live = true;
	return true;
}

// ---- small-call regime: see the member comment.  Chosen when the calls are at most an eighth of the filter.
bool ConvStage::init_fdl(const Spec &sp, ssize_t max_frames)
{
	static const int env = [] { const char *e = getenv("DSP_AMD_CONV_FDL"); return e ? atoi(e) : -1; }();   // 0 = never, P1 = force that many head partitions
	// (`fir` / the zita contract: every window lat frames earlier, head and tail alike; the zita contract's float32 roundings are those of
	// its inputs -- already in the rings --, of its taps and of the finished output: transforms and delay lines stay fp64 here)
	if (env == 0 || resampler || nph != 1 || merged_pre) return true;
	if (max_frames < 256 || (long) max_frames * 8 > T) return true;
	long b = 2048;
	while (b >= 256 && (max_frames % b)) b >>= 1;
	if (b < 256) return true;
	fB = b; fNF = 2 * b;
	const long parts = (T + fB - 1) / fB;
	static const bool upc_on = [] { const char *e = getenv("DSP_AMD_CONV_UPC"); return !e || atoi(e) != 0; }();
	auto upc_tail_ok = [&](long D) { return upc_on && T - D > D && 2 * D >= (1L << (FFT_MIN_LOG2_N2 + FFT_MIN_LOG2_N1)) && 2 * D <= (1L << (FFT_MAX_LOG2_N2 + FFT_MAX_LOG2_N1)); };
	if (env > 0) fP1 = (int) std::min<long>(parts, env);
	else if (parts <= 16) fP1 = (int) parts;                   // short enough: the whole filter in the delay line, no tail
	else {
		// head partitions against the tail's work, in units of 32 bytes per pair and frame: the head moves P1 + 2 of them (at about two
		// thirds of the rate the four-step kernels reach: every workgroup of conv_fdl walks through the same phases at the same time),
		// a delay-line tail on blocks of D = P1 B frames its slots + 5.5, a one-transform tail 3.25 N / D
		double best = 0.0;
		for (int c : { 4, 8, 16 }) {
			const long D = (long) c * fB;
			const double tail = upc_tail_ok(D) ? (double) ((T - D + D - 1) / D) + 5.5
			                                   : 3.25 * (double) std::max<long>(next_pow2(T - D + D), 1L << (FFT_MIN_LOG2_N2 + FFT_MIN_LOG2_N1)) / (double) D;
			const double cost = 1.5 * (c + 2) + tail;
			if (best == 0.0 || cost < best) { best = cost; fP1 = c; }
		}
	}
	fD = (long) fP1 * fB;
	const long n_pairs = (long) S * pps;
	// conv_fdl addresses a pair's delay-line slots and its ring row with 32-bit byte offsets (buffer descriptors)
	if ((double) fP1 * n_pairs * fNF * sizeof(double2) >= 2.0e9 || (double) ring_len * sizeof(double2) >= 2.0e9) return true;   // (regime off: one transform per call)
	if (!fdl_buf.alloc((size_t) fP1 * n_pairs * fNF * sizeof(double2)) || !fdl_H.alloc((size_t) n_filters * fP1 * fNF * sizeof(double2), false)) return false;
	std::vector<double2> t;
	make_twiddles(fNF, fNF, 1, t);
	if (!fdl_tw.upload(t.data(), t.size() * sizeof(double2))) return false;
	{
		// head partition spectra: partition q = taps [q B, q B + B), zero-padded to 2 B, through the kernel's own forward transform
		// (one set of rows per filter: a shared filter, or one per selected channel, fir_p.c:483-495)
		std::vector<double2> rows((size_t) n_filters * fP1 * fNF, make_double2(0.0, 0.0));
		for (int f = 0; f < n_filters; ++f)
			for (int q = 0; q < fP1; ++q)
				for (long i = 0; i < fB && q * fB + i < T; ++i) {
					const double t = sp.taps[(size_t) (q * fB + i) * sp.fch + (sp.fch == 1 ? 0 : f)];
					rows[((size_t) f * fP1 + q) * fNF + i].x = round_f32 ? (double) (float) t : t;
				}
		DevBuf d_rows;
		if (!d_rows.upload(rows.data(), rows.size() * sizeof(double2))) return false;
		FdlParams fp;
		memset(&fp, 0, sizeof(fp));
		fp.log2NF = ilog2(fNF); fp.P1 = fP1; fp.NF = fNF; fp.B = fB;
		fp.ring = d_rows.as<double2>(); fp.ring_row_stride = fNF; fp.ring_mask = fNF - 1; fp.win_base = 0;
		fp.n_sub = 1;
		fp.spec_out = fdl_H.as<double2>(); fp.h_scale = 1.0 / (double) fNF;
		fp.tw_nf = fdl_tw.as<double2>();
		fp.n_pairs = (long) n_filters * fP1; fp.C = ch_in; fp.pairs_per_stream = 1;
		launch_conv_fdl(fp, nullptr);
		if (!hip_ok(hipDeviceSynchronize(), "head partition spectra")) return false;
	}
	if (fD < T) {
		Spec ts(sp);
		ts.T = T - fD;
		ts.taps.assign(sp.taps.begin() + (size_t) fD * sp.fch, sp.taps.end());
		ts.name = sp.name + ":tail";
		if (round_f32) {
			// the tail's share is added to the head's in fp64 and rounded once, by conv_fdl: the child is a plain fp64 stage on float32 taps
			ts.conv_mode = CONV_LATENCY_LEN;
			for (double &t : ts.taps) t = (double) (float) t;
		}
		tail_conv.reset(new ConvStage);
		tail_conv->S = S; tail_conv->ch_in = ch_in; tail_conv->ch_out = ch_out; tail_conv->fs_in = fs_in; tail_conv->fs_out = fs_out;
		const long fn = (ts.T - 1 + 7) & ~7L;
		// more than one block of taps left: partitions of fD taps with a delay line of their own (transforms of 2 fD points instead of
		// one of >= T - fD + fD points per fD frames); else the plain overlap-save tail
		const bool upc_tail = upc_tail_ok(fD);
		if (!tail_conv->init(ts, fD, nullptr, nullptr, this, upc_tail ? 0 : std::max<long>(next_pow2(fn + fD), 1L << (FFT_MIN_LOG2_N2 + FFT_MIN_LOG2_N1)), upc_tail ? fD : 0)
		    || !tail_buf.alloc((size_t) S * fD * ch_in * sizeof(double))) {
			// the regime is an optimisation: without it the stage runs one transform per call as before
			log_msg(LL_VERBOSE, "%s: info: small-call regime not available (%s)", name.c_str(), last_error());
			tail_conv.reset();
			fdl_buf.release(); fdl_H.release(); fdl_tw.release(); tail_buf.release();
			return true;
		}
	}
	fdl = fdl_live = true;
	return true;
}

// the call's frames are already in the rings (pos .. pos + frames); frames and q_abs are multiples of fB
void ConvStage::run_fdl(ssize_t frames, double *out, long out_stride, hipStream_t st)
{
	const long n = frames / fB;
	long done = 0;
	while (done < n) {
		const long q_now = q_abs + done * fB;
		const long seg = std::min<long>(n - done, (fD - q_now % fD) / fB);
		FdlParams fp;
		memset(&fp, 0, sizeof(fp));
		fp.log2NF = ilog2(fNF); fp.P1 = fP1; fp.NF = fNF; fp.B = fB;
		fp.ring = ring_dev; fp.ring_row_stride = ring_stride; fp.ring_mask = ring_len - 1;
		fp.win_base = (pos + done * fB - fB - lat) & (ring_len - 1);
		fp.n_sub = (int) seg;
		fp.slot0 = f_slot;
		fp.fdl = fdl_buf.as<double2>();
		fp.Hf = fdl_H.as<double2>();
		fp.pair_h = (n_filters > 1) ? pair_h.as<int>() : nullptr;
		fp.tw_nf = fdl_tw.as<double2>();
		fp.n_pairs = (long) S * pps;
		fp.C = ch_in; fp.pairs_per_stream = pps;
		fp.pair_out_ch = pair_out_ch.as<int>();
		fp.out = out + (size_t) done * fB * ch_in;
		fp.out_stride_frames = out_stride;
		fp.round_f32 = round_f32;
		fp.sink = wire_sink;
		if (wire_sink.on) {
			// (`out` holds samples of the sink's format; the dither sequence goes on where the sub-blocks before this launch left it)
			fp.out = reinterpret_cast<double *>(reinterpret_cast<char *>(out) + (size_t) done * fB * ch_in * pcm_sample_bytes(wire_sink.fmt));
			fp.sink.samples_before += done * fB * ch_in;
		}
		if (tail_conv) { fp.tail = tail_buf.as<double>(); fp.tail_stride_frames = fD; fp.tail_off = q_now % fD; }
		{ ProfScope ps("conv_fdl", st); launch_conv_fdl(fp, st); }
		f_slot = (int) ((f_slot + seg) % fP1);
		done += seg;
		const long n0 = q_abs + done * fB;
		if (tail_conv && n0 % fD == 0) {
			// the tail's share of the NEXT fD outputs: (h_tail * x)[n0 - fD .. n0), every input it needs is in the rings by now
			tail_conv->convolve(n0 - fD, n0 - 1, n0 - fD, fD, tail_buf.as<double>(), fD, st);
		}
	}
}

// filter spectra: run the forward half of the pipeline on the taps themselves (exactly what the
// reference does at init with its r2c plan, fir.c:342-357 / fir_p.c:482-498)
bool ConvStage::spectrum_of(const std::vector<double> &src, long n_taps, int stride, int offset, double2 *dst, int row_nph)
{
	DevBuf tring, tph;
	std::vector<double2> taps(N, make_double2(0.0, 0.0));
	std::vector<int> hsel{ 0 };
	if (!tring.alloc((size_t) N * sizeof(double2), false)) return false;
	if (!tph.upload(hsel.data(), hsel.size() * sizeof(int))) return false;
	for (long i = 0; i < n_taps; ++i) {
		double v = src[(size_t) i * stride + offset];
		if (round_f32) v = (double) (float) v;
		taps[i].x = v;
	}
	if (!hip_ok(hipMemcpy(tring.p, taps.data(), (size_t) N * sizeof(double2), hipMemcpyHostToDevice), "H2D taps")) return false;
	ConvParams p = base_params();
	p.f32 = 0;                                   // always the fp64 kernels and tables (a float32 stage rounds the result: spectrum_f32)
	p.tw_n1 = tw_n1.as<double2>(); p.tw_n2 = tw_n2.as<double2>(); p.tw_col = tw_col.as<double2>();
	p.ring = tring.as<double2>();
	p.ring_row_stride = N; p.ring_mask = N - 1;
	p.win_base = 0; p.valid = n_taps;
	p.pair_h = tph.as<int>();
	p.pair0 = 0;
	p.Hout = dst;
	p.nph = row_nph;   // selects the row kernel whose order H is stored in
	launch_conv_col(p, false, 1, nullptr);
	launch_conv_row(p, 1, 1, nullptr);
	return hip_ok(hipDeviceSynchronize(), "filter spectrum");
}

// filter spectrum number `index` of a float32 stage: computed in fp64, rounded once to float2
bool ConvStage::spectrum_f32(const std::vector<double> &src, long n_taps, int stride, int offset, size_t index, int row_nph)
{
	DevBuf hd;
	if (!hd.alloc((size_t) N * sizeof(double2), false) || !spectrum_of(src, n_taps, stride, offset, hd.as<double2>(), row_nph)) return false;
	std::vector<double2> h((size_t) N);
	if (!hip_ok(hipMemcpy(h.data(), hd.p, (size_t) N * sizeof(double2), hipMemcpyDeviceToHost), "D2H spectrum")) return false;
	std::vector<float2> hf((size_t) N);
	for (long i = 0; i < N; ++i) hf[(size_t) i] = make_float2((float) h[(size_t) i].x, (float) h[(size_t) i].y);
	return hip_ok(hipMemcpy(static_cast<char *>(H.p) + index * (size_t) N * sizeof(float2), hf.data(), (size_t) N * sizeof(float2), hipMemcpyHostToDevice), "H2D spectrum");
}

// filter spectra of the one-trip form: the kernel's own forward transform of the taps as a window of N points (preparation mode), natural order
bool ConvStage::prepare_short(const Spec &sp)
{
	// (uniformly partitioned form: row f upc_P + q = taps [q upc_B, q upc_B + upc_B) of filter f, fir_p.c:483-495)
	const int parts = upc_P ? upc_P : 1;
	const long part_len = upc_P ? upc_B : T;
	std::vector<double2> rows((size_t) n_filters * parts * N, make_double2(0.0, 0.0));
	for (int f = 0; f < n_filters; ++f)
		for (int q = 0; q < parts; ++q)
			for (long i = 0; i < part_len && q * part_len + i < T_taps; ++i)
				rows[((size_t) f * parts + q) * N + i].x = sp.taps[(size_t) (q * part_len + i) * sp.fch + (sp.fch == 1 ? 0 : f)];
	DevBuf d_rows;
	if (!d_rows.upload(rows.data(), rows.size() * sizeof(double2))) return false;
	ShortParams p;
	memset(&p, 0, sizeof(p));
	p.N = N; p.first_n = 0; p.hop = N;
	p.ring = d_rows.as<double2>(); p.ring_row_stride = N; p.ring_mask = N - 1;
	p.q0 = 0; p.lat = 0; p.n_in = N;
	p.C = 2; p.pairs_per_stream = 1;
	p.pair_out_ch = pair_out_ch.as<int>(); p.pair_h = pair_h.as<int>();
	p.Hout = H.as<double2>(); p.h_scale = 1.0 / (double) N;
	p.tw = tw_short.as<double2>();
	p.slab_fmt = PCM_DOUBLE;
	p.n_pairs = (long) n_filters * parts; p.blocks_per_wg = 1;
	launch_conv_short(p, nullptr);
	return hip_ok(hipDeviceSynchronize(), "filter spectrum");
}

void ConvStage::convolve_short(long q_lo, long q_hi, long k_origin, long out_count, double *out, long out_stride, hipStream_t st)
{
	ShortParams p;
	memset(&p, 0, sizeof(p));
	p.N = N; p.first_n = first_n; p.hop = B;
	p.ring = ring_dev; p.ring_row_stride = ring_stride; p.ring_mask = ring_len - 1;
	p.q0 = q_lo; p.lat = lat; p.n_in = q_hi - q_lo + 1;
	p.slab_fmt = PCM_DOUBLE;
	if (cur_slab) { p.slab = cur_slab; p.slab_stride_frames = cur_slab_stride; p.slab_frames = cur_frames; p.slab_q0 = cur_q0; p.file_from = cur_q0 + cur_frames - first_n; p.slab_fmt = wire_in_fmt; }
	p.sink = wire_sink;
	p.C = ch_in; p.pairs_per_stream = pps;
	p.pair_out_ch = pair_out_ch.as<int>(); p.pair_h = pair_h.as<int>();
	p.H = H.as<double2>(); p.tw = tw_short.as<double2>();
	p.out = out; p.out_stride_frames = out_stride; p.k_origin = k_origin; p.out_count = out_count;
	p.ring_out = feed_ring; p.ring_out_stride = feed_stride; p.ring_out_mask = feed_mask; p.ring_out_pos = feed_pos; p.ring_out_round_f32 = feed_round;
	p.round_f32 = round_f32;
	p.n_pairs = (long) S * pps;
	// a workgroup walks one pair's blocks; few pairs: the blocks of a pair are shared out until two workgroups per CU's worth exist
	const long n_blocks = (p.n_in + B - 1) / B;
	long ranges = (512 + p.n_pairs - 1) / p.n_pairs;
	ranges = std::max<long>(1, std::min<long>(ranges, n_blocks));
	if (upc_P) {
		// (the delay line: every block reads what the blocks before it wrote -- one workgroup per pair, in order)
		p.fdl = upc_buf.as<double2>(); p.fdl_slot_stride = (long) S * pps * N; p.fdl_P = upc_P; p.fdl_slot = upc_slot;
		upc_slot = (int) ((upc_slot + n_blocks) % upc_P);
		ranges = 1;
	}
	p.blocks_per_wg = (int) ((n_blocks + ranges - 1) / ranges);
	ProfScope ps("conv_short", st);
	launch_conv_short(p, st);
}

bool ConvStage::prepare_filters(const Spec &sp)
{
	if (short_mode) return prepare_short(sp);
	if (upc_P) {
		// partition q = taps [q B, q B + B) of filter f (one shared filter, or one per selected channel: fir_p.c:483-495), zero-padded to the transform
		for (int f = 0; f < n_filters; ++f)
			for (int q = 0; q < upc_P; ++q) {
				const long lo = (long) q * upc_B, n = std::min<long>(upc_B, T_taps - lo);
				std::vector<double> part((size_t) n);
				for (long i = 0; i < n; ++i) part[(size_t) i] = sp.taps[(size_t) (lo + i) * sp.fch + (sp.fch == 1 ? 0 : f)];
				const size_t idx = (size_t) f * upc_P + q;
				if (!(f32 ? spectrum_f32(part, n, 1, 0, idx, 1) : spectrum_of(part, n, 1, 0, H.as<double2>() + idx * N, 1))) return false;
			}
		return true;
	}
	for (int f = 0; f < n_filters * nph; ++f) {
		bool ok;
		if (f32) ok = resampler ? spectrum_f32(rs_tab, T, up, f, (size_t) f, nph) : spectrum_f32(sp.taps, T, sp.fch, f, (size_t) f, nph);
		else ok = resampler ? spectrum_of(rs_tab, T, up, f, H.as<double2>() + (size_t) f * N, nph)
		                    : spectrum_of(sp.taps, T, sp.fch, f, H.as<double2>() + (size_t) f * N, nph);
		if (!ok) return false;
	}
	if (merged_pre) {
		if (!H_plain.alloc((size_t) N * sizeof(double2), false)) return false;
		// H_plain is used by the single-phase row kernel (nph = 1 selects it and its H order)
		if (!spectrum_of(pre_taps, (long) pre_taps.size(), 1, 0, H_plain.as<double2>(), 1)) return false;
		if (!tail_z.alloc((size_t) S * (J_rs + 8) * ch_in * sizeof(double))) return false;
		if (!tail_scratch.alloc((size_t) S * ((size_t) (J_rs + 8) * up + 8) * ch_in * sizeof(double))) return false;
		if (!tail_out.alloc((size_t) S * (out_delay + 8) * ch_in * sizeof(double))) return false;
	}
	return true;
}

void ConvStage::push(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st)
{
	DeintParams d;
	d.in = in;
	d.out = all_selected ? nullptr : out;
	d.in_stride_frames = in_stride; d.out_stride_frames = out_stride; d.frames = frames;
	d.C = ch_in;
	d.slot_of_channel = slot_of_channel.as<int>();
	d.rows_per_stream = pps;
	d.ring = ring_dev;
	d.ring_row_stride = ring_stride; d.ring_mask = ring_len - 1; d.pos = pos;
	d.round_f32 = round_f32;
	d.in_fmt = wire_in_fmt;
	{ ProfScope ps("conv_deinterleave", st); launch_deinterleave(d, S, st); }
}

// Convolution outputs at the absolute input indices q_lo .. q_hi (the ring holds input index q at q & mask; indices
// at or beyond q_end read as zero: the drain), mapped to output frames by (up, down, k_origin): see fft_params.h.
void ConvStage::convolve(long q_lo, long q_hi, long k_origin, long out_count, double *out, long out_stride, hipStream_t st)
{
	if (short_mode) { convolve_short(q_lo, q_hi, k_origin, out_count, out, out_stride, st); return; }
	const long q_end = resampler ? q_total : q_hi + 1;
	for (long q_blk = q_lo; q_blk <= q_hi; q_blk += B) {
		const long f = std::min<long>(B, q_hi - q_blk + 1);
		ConvParams p = base_params();
		p.win_base = (q_blk - lat - first_n) & (ring_len - 1);   // two's complement wrap: ring_len is a power of two
		p.first_n = first_n;
		p.valid = std::max<long>(0, std::min<long>(first_n + f, first_n + q_end - q_blk));
		p.out = out;
		p.out_stride_frames = out_stride;
		p.sink = wire_sink;
		p.in_count = f;
		p.q_blk = q_blk;
		p.k_origin = k_origin;
		p.out_count = out_count;
		p.k3_pipe_ok = (all_selected && n_filters == 1 && pps == 4 && ch_in == 8 && !feeds && ((((size_t) out) & 15) == 0)) ? 1 : 0;
		if (cur_slab) { p.slab = cur_slab; p.slab_stride_frames = cur_slab_stride; p.slab_frame0 = q_blk - cur_q0; p.slab_store = resampler ? 0 : 1; p.slab_fmt = wire_in_fmt; }
		const int row_mode = upc_P ? 3 : (nph > 1) ? 2 : 0;
		if (upc_P) {
			p.fdl = upc_buf.as<double2>(); p.fdl_slot_stride = (long) S * pps * N; p.fdl_P = upc_P; p.fdl_slot = upc_slot;
			upc_slot = (upc_slot + 1) % upc_P;
		}
		// every stream in one launch of each kernel
		const int n_pairs = (int) ((long) S * pps);
		p.pair0 = 0;
		p.stream0 = 0;
		p.n_streams_launch = S;
		if (fuse_this_call) {
			// (a resampler's whole-window call behind a cascade: the fused first pass in K1's place; one block)
			fuse_this_call = false;
			if (q_blk != q_lo || f != B || p.valid != N || !fused_first_pass(p, f, st)) { fuse_failed = true; return; }
		}
		else { ProfScope ps("conv_col_fwd", st); launch_conv_col(p, false, n_pairs, st); }
		{ ProfScope ps("conv_row", st); launch_conv_row(p, row_mode, n_pairs, st); }
		{ ProfScope ps("conv_col_inv", st); launch_conv_col(p, true, n_pairs, st); }
	}
}

// resampler: visible output frames [emitted, emitted + count) = full-rate indices k' = emitted + out_delay + ...,
// each the convolution at input index floor(k' down / up)
ssize_t ConvStage::emit(long count, double *out, long out_stride, hipStream_t st)
{
	if (count <= 0) return 0;
	const long kp0 = emitted + out_delay, kp1 = kp0 + count;
	convolve((kp0 * down) / up, ((kp1 - 1) * down) / up, kp0, count, out, out_stride, st);
	emitted += count;
	feed_pos = (feed_pos + count) & feed_mask;
	return count;
}

ssize_t ConvStage::run(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st)
{
	if (upc_conv && upc_live) {
		if (!feeds && frames % upc_conv->upc_B == 0 && q_abs % upc_conv->upc_B == 0) {
			cur_slab = nullptr;
			if (!fed) push(in, in_stride, frames, out, out_stride, st);
			upc_conv->wire_sink = wire_sink;                 // (a plain convolution of every channel or not: K3's stores are the same kernel's)
			// (leading frames to drop -- absorb_discard: `fir`'s latency -- still go through the delay line, block by block; K3 starts writing later)
			const long drop = std::min<long>(skip_left, frames);
			upc_conv->convolve(q_abs, q_abs + frames - 1, q_abs + drop, frames - drop, out, out_stride, st);
			skip_left -= drop;
			q_abs += frames;
			pos = (pos + frames) & (ring_len - 1);
			return frames - drop;
		}
		upc_live = false;
	}
	if (fdl && fdl_live) {
		if (!feeds && frames % fB == 0 && q_abs % fB == 0) {
			cur_slab = nullptr;
			if (!fed) push(in, in_stride, frames, out, out_stride, st);
			run_fdl(frames, out, out_stride, st);
			q_abs += frames;
			pos = (pos + frames) & (ring_len - 1);
			return frames;
		}
		fdl_live = false;      // off the grid: the rings carry on with one transform per call (the delay line is a cache of them)
	}
	const bool use_direct = direct && !fed && ((((size_t) in) & 15) == 0) && (double) in_stride * ch_in * 8 < 1.0e18;
	cur_slab = use_direct ? in : nullptr;
	cur_slab_stride = in_stride;
	cur_frames = frames;
	cur_q0 = resampler ? q_total : q_abs;
	if (!fed && !use_direct) push(in, in_stride, frames, out, out_stride, st);
	if (resampler) {
		const long pos0 = pos;
		pos = (pos + frames) & (ring_len - 1);
		q_total += frames;
		// full-rate outputs k with floor(k down / up) < q_total are computable: k < ceil(q_total up / down)
		const long avail = std::max<long>(0, max_out_frames(q_total) - out_delay) - emitted;
		// (the feeding cascade left this call to the fused kernels: fuse_accepts said yes to exactly this call -- one whole window)
		fuse_this_call = feeder_ && feeder_->pending.in;
		fuse_failed = false;
		const ssize_t got = emit(std::min<long>(avail, max_out_frames(frames)), out, out_stride, st);
		if (fuse_this_call || fuse_failed) { fuse_this_call = false; set_error("%s: fused first pass could not be launched", name.c_str()); return PIPE_FAILED; }
		if (use_direct) {
			// the windows of later calls (and of the drain) start at the next unemitted output's input index minus the history:
			// only that tail of this call has to be in the ring
			long keep_from = ((emitted + out_delay) * down) / up - first_n - 8;
			if (merged_pre) keep_from = std::min(keep_from, q_total - (J_rs + down + first_n + 16));
			const long off = std::max<long>(0, std::min<long>(frames, keep_from - cur_q0));
			if (off < frames) {
				const long sv = pos;
				pos = (pos0 + off) & (ring_len - 1);
				push(reinterpret_cast<const double *>(reinterpret_cast<const char *>(in) + (size_t) off * ch_in * pcm_sample_bytes(wire_in_fmt)), in_stride, frames - off, out, out_stride, st);
				pos = sv;
			}
			cur_slab = nullptr;
		}
		return got;
	}
	if (feeder_ && feeder_->pending.in) {
		// (the feeding cascade left this call to the fused kernels: fuse_accepts said yes to exactly this call)
		if (!run_fused(frames, out, out_stride, st)) { set_error("%s: fused first pass could not be launched", name.c_str()); return PIPE_FAILED; }
		q_abs += frames;
		feed_pos = (feed_pos + frames) & feed_mask;
		pos = (pos + frames) & (ring_len - 1);
		return frames;
	}
	if (fuse_plain && use_direct && wire_in_fmt == PCM_DOUBLE && frames == B && skip_left == 0 && !feeds && (q_abs & 7) == 0) {
		// (one whole hop straight from the slab: the first pass in its two-pairs-per-workgroup form; it files the rows the next window looks back at)
		if (!run_fused_plain(in, in_stride, frames, out, out_stride, st)) { set_error("%s: fused first pass could not be launched", name.c_str()); return PIPE_FAILED; }
		cur_slab = nullptr;
		q_abs += frames;
		feed_pos = (feed_pos + frames) & feed_mask;
		pos = (pos + frames) & (ring_len - 1);
		return frames;
	}
	// plain convolution: output frame m of this call = convolution at ring index pos + m
	const long drop = std::min<long>(skip_left, frames);
	if (drop < frames) convolve(q_abs + drop, q_abs + frames - 1, q_abs + drop, frames - drop, out, out_stride, st);
	skip_left -= drop;
	q_abs += frames;
	feed_pos = (feed_pos + frames - drop) & feed_mask;     // (a consumer fed through its ring sees the frames that were written)
	pos = (pos + frames) & (ring_len - 1);
	return frames - drop;
}

ssize_t ConvStage::drain2(ssize_t max_frames, double *out, long out_stride, hipStream_t st)
{
	if (!resampler) return -1;
	cur_slab = nullptr;
	// total output length is ceil(N up / down) (resample.c:163-188): the tail is computed against zero input
	const long left = max_out_frames(q_total) - emitted;
	if (q_total == 0 || left <= 0) return -1;
	const long count = std::min<long>(left, std::max<long>(max_out_frames(max_frames), 1));
	if (!merged_pre) return emit(count, out, out_stride, st);
	if (tail_frames < 0 && !compute_tail(st)) return PIPE_FAILED;
	// serve the precomputed tail
	const long n = std::min<long>(count, tail_frames - tail_served);
	if (n <= 0) return -1;
	launch_copy_slab(tail_out.as<double>() + (size_t) tail_served * ch_in, out_delay + 8, out, out_stride, n, 0, ch_in, S, st);
	tail_served += n;
	emitted += n;
	return n;
}

// the drain2 tail of a merged fir_p + resample stage, the reference's way (see the member comment)
bool ConvStage::compute_tail(hipStream_t st)
{
	// 1. z = fir_p output at the last Jt >= J input indices [q_total - Jt, q_total): plain filter, single phase, into
	//    tail_z.  The helper's stream starts at q_total - Jt: a multiple of `down`, so that its decimation phase is
	//    the stream's.
	const long Jt = J_rs + (((q_total - J_rs) % down) + down) % down;
	{
		const int sv_nph = nph, sv_up = up, sv_down = down;
		std::swap(H.p, H_plain.p); std::swap(H.bytes, H_plain.bytes);
		nph = 1; up = 1; down = 1;
		// inputs at or beyond q_total read as zero, as in the resampler path
		convolve(q_total - Jt, q_total - 1, q_total - Jt, Jt, tail_z.as<double>(), J_rs + 8, st);
		nph = sv_nph; up = sv_up; down = sv_down;
		std::swap(H.p, H_plain.p); std::swap(H.bytes, H_plain.bytes);
	}
	// 2. a polyphase resampler of its own over those frames: its regular output is not needed, its drain2 is the tail
	tail_rs->reset(st);
	const long reg = tail_rs->max_out_frames(Jt);
	if (tail_rs->run(tail_z.as<double>(), J_rs + 8, Jt, tail_scratch.as<double>(), reg + 8, st) < 0) return false;
	tail_frames = 0;
	for (;;) {
		const ssize_t got = tail_rs->drain2(J_rs, tail_out.as<double>() + (size_t) tail_frames * ch_in, out_delay + 8, st);
		if (got <= 0) break;
		tail_frames += got;
		if (tail_frames >= out_delay) break;
	}
	tail_served = 0;
	return true;
}

void ConvStage::reset(hipStream_t st)
{
	if (ring.p) (void) hipMemsetAsync(ring.p, 0, ring.bytes, st);
	pos = 0;
	q_total = emitted = q_abs = 0;
	skip_left = skip;
	if (upc_P) { (void) hipMemsetAsync(upc_buf.p, 0, upc_buf.bytes, st); upc_slot = 0; }
	tail_frames = -1; tail_served = 0;
	feed_pos = 0;
	if (feeder_) feeder_->ring.pos = 0;
	if (upc_conv) { upc_conv->reset(st); upc_live = true; }
	if (fdl) {
		(void) hipMemsetAsync(fdl_buf.p, 0, fdl_buf.bytes, st);
		if (tail_buf.p) (void) hipMemsetAsync(tail_buf.p, 0, tail_buf.bytes, st);
		if (tail_conv) tail_conv->reset(st);            // (its delay lines; the rings it works on are this stage's)
		f_slot = 0;
		fdl_live = true;
	}
}

// -------------------------------------------------------------- FirDirectStage

class FirDirectStage : public Stage {
public:
	bool init(const Spec &sp);
	const char *type() const override { return "fir_direct"; }
	std::string describe() const override { return "fir_direct[" + name + " T=" + std::to_string(T) + "]"; }
	ssize_t run(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st) override;
	void reset(hipStream_t st) override { (void) hipMemsetAsync(hist.p, 0, hist.bytes, st); phase = 0; }
	size_t device_bytes() const override { return hist.bytes + taps.bytes; }
	friend bool fir_direct_view(Stage *s, ResidentPass *ps, const int **phase);
private:
	int T = 0, phase = 0;
	std::string name;
	DevBuf taps, foc, hist;
};

bool fir_direct_view(Stage *s, ResidentPass *ps, const int **phase)
{
	FirDirectStage *f = dynamic_cast<FirDirectStage *>(s);
	if (!f || f->S != 1 || f->ch_in > RES_FIR_MAX_CH || f->T < 1 || f->T > RES_FIR_TAPS) return false;
	static_assert(RES_FIR_TAPS == FIR_DIRECT_MAX, "the wave reads the stage's own tables");
	memset(ps, 0, sizeof(*ps));
	ps->kind = RES_PASS_FIR; ps->c_in = ps->c_out = f->ch_in;
	ps->T = f->T; ps->taps = f->taps.as<double>(); ps->foc = f->foc.as<int>(); ps->hist = f->hist.as<double>();
	*phase = &f->phase;
	return true;
}

bool FirDirectStage::init(const Spec &sp)
{
	name = sp.name;
	T = (int) sp.T;
	if (T > FIR_DIRECT_MAX) { set_error("%s: BUG: direct FIR longer than %d taps", name.c_str(), FIR_DIRECT_MAX); return false; }
	const int nf = sp.fch;
	std::vector<double> h((size_t) nf * FIR_DIRECT_MAX, 0.0);
	for (int f = 0; f < nf; ++f) for (int i = 0; i < T; ++i) h[(size_t) f * FIR_DIRECT_MAX + i] = sp.taps[(size_t) i * nf + f];
	std::vector<int> fo(ch_in, -1);
	for (int c = 0, k = 0; c < ch_in; ++c) if (sp.sel[c]) { fo[c] = (nf == 1) ? 0 : k; ++k; }
	if (!taps.upload(h.data(), h.size() * sizeof(double))) return false;
	if (!foc.upload(fo.data(), fo.size() * sizeof(int))) return false;
	return hist.alloc((size_t) 2 * S * ch_in * FIR_DIRECT_MAX * sizeof(double));
}

ssize_t FirDirectStage::run(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st)
{
	const size_t half = (size_t) S * ch_in * FIR_DIRECT_MAX;
	FirDirectParams p;
	p.in = in; p.out = out;
	p.in_stride_frames = in_stride; p.out_stride_frames = out_stride; p.frames = frames;
	p.C = ch_in; p.T = T;
	p.filter_of_channel = foc.as<int>();
	p.taps = taps.as<double>();
	p.hist_rd = hist.as<double>() + (phase ? half : 0);
	p.hist_wr = hist.as<double>() + (phase ? 0 : half);
	{ ProfScope ps("fir_direct_kernel", st); launch_fir_direct(p, S, st); }
	phase ^= 1;
	return frames;
}

// ------------------------------------------------------------------ factory

Stage *make_conv_stage(const Spec &sp, int n_streams, ssize_t max_frames, CascadeStage *feeder, Stage *prev)
{
	// integer ratios ride the FFT convolver (one forward transform, one inverse per polyphase branch); general n/d
	// stays on the polyphase dot-product kernel
	if (sp.kind == Kind::Resample && !((sp.rs_n == 1 || sp.rs_d == 1) && sp.rs_n <= 8 && sp.rs_d <= 8 && !getenv("DSP_AMD_RESAMPLE_DIRECT")))
		return make_resample_stage(sp, n_streams, max_frames);
	if (sp.kind == Kind::FirDirect) {
		FirDirectStage *s = new FirDirectStage;
		s->S = n_streams; s->ch_in = sp.ch_in; s->ch_out = sp.ch_out; s->fs_in = sp.fs_in; s->fs_out = sp.fs_out;
		if (!s->init(sp)) { delete s; return nullptr; }
		return s;
	}
	ConvStage *s = new ConvStage;
	s->S = n_streams; s->ch_in = sp.ch_in; s->ch_out = sp.ch_out; s->fs_in = sp.fs_in; s->fs_out = sp.fs_out;
	if (!s->init(sp, max_frames, feeder, prev)) { delete s; return nullptr; }
	return s;
}

}  // namespace dspamd
