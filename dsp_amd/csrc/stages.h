// stages.h -- concrete Stage classes
#pragma once
#include "engine.h"
#include <cstdlib>
#include <functional>

namespace dspamd {

// DSP_AMD_NO_WIRE_FUSION: the wire formats always as passes of their own (the fallback suite's switch)
inline bool wire_fusion_on() { static const bool off = getenv("DSP_AMD_NO_WIRE_FUSION") != nullptr; return !off; }

// gain / add / biquad sections on the same stream format, fused into one launch
class CascadeStage : public Stage {
public:
	void add(const Spec &sp);
	bool finalize();
	const char *type() const override { return "cascade"; }
	std::string describe() const override;
	bool in_place_ok() const override { return true; }
	ssize_t run(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st) override;
	void reset(hipStream_t st) override;
	bool wire_in_ok(int fmt, const void *in, long in_stride, ssize_t frames, bool also_out, int out_fmt) const override;
	bool wire_out_ok(int fmt, const void *out, long out_stride, ssize_t frames, bool also_in, int in_fmt) const override;
	size_t device_bytes() const override { return ops.bytes + state.bytes + fops.bytes + fq.bytes; }
	// optional planar destination owned by the following FFT convolver
	PlanarRing ring = { nullptr, 0, 0, 0, nullptr, 0, 0 };
	int write_interleaved = 1;
	int n_ops = 0, Cg = 1;
	// ---- the cascade fused into the following convolver's first pass (kernels_fused.hip; ConvStage::run_fused) ----
	// The convolver installs fuse_probe; a call it accepts is not run here at all: run() only notes where the call's frames lie
	// (`pending`) and the convolver's run() does the whole of it -- end states of the window's rows from zero state, the scan over
	// them (the chunk plan's tables), the recurrence from the true states in front of the column transforms.  State and rings
	// are left exactly as the separate kernels leave them, so any later call may take the ordinary path again.
	std::function<bool(const void *in, long in_stride, ssize_t frames, int in_fmt)> fuse_probe;     // in_fmt: PCM_DOUBLE or the wire format `in` holds
	struct Pending { const double *in = nullptr; long in_stride = 0; ssize_t frames = 0; int in_fmt = PCM_DOUBLE; } pending;
	// sections of a chain whose channels all run the same sections and gains (gains folded into the next section's b coefficients,
	// as cascade_rows has them), padded with pass-through sections to a count the fused kernels are instantiated for
	// (round 5: or whose channel PAIRS each run their own sections -- the two channels of a pair alike, `pairs` tables then: sections behind a
	// pair-aligned selector are pass-through sections in the other pairs' tables, gains are per pair)
	struct FuseTables { bool tried = false, ok = false; int n_sec = 0, n_real = 0, pairs = 1; double gain = 1.0; DevBuf sec, sec_op, gain_tab; };   // n_sec: with the padding, n_real: the chain's own; pairs: 1 = one table for every pair
	const FuseTables &fuse_tables();
	// the device tables a resident small-block wave works on (kernels_resident.hip): [C][n_ops] ops, [S][C][n_ops][2] states; sections per channel at most
	const OpDesc *device_ops() const { return ops.as<OpDesc>(); }
	double *device_state() { return state.as<double>(); }

	friend class ConvStage;
private:
	FuseTables fuse_tab;
	std::vector<std::vector<OpDesc>> cols;   // [op][channel]
	std::vector<std::string> names;
	DevBuf ops, state, fops, fq, frows, frq;
	bool rows4_ok = false;
	// few channels, long calls: the time axis is cut into K chunks that run as independent zero-state "streams" (kernels_chunk.hip)
	std::vector<OpDesc> host_ops;            // [C][n_ops]
	bool chunk_linear = false;               // sections and gains only (an `add` is not linear in the state)
	// Whether a call runs chunked is a pure function of (frames, S, the sections) -- never of earlier calls: run / reset / run
	// and two instances with different histories give bit-identical output (the chunked path agrees with the direct kernels
	// only to rounding, ~1e-15).  Plans (tables + buffers, about a millisecond each) are kept for the last few call sizes.
	struct ChunkPlan {
		long frames = 0, len = 0; int K = 0, n_pow = 0, n_cls = 0; DevBuf cls, H, Mp, cstate, X;
		DevBuf G; int g_states = 0;                            // fused path: [len][32] input-to-end-state table of the matrix-core prepass (fuse_gtable)
		hipEvent_t done = nullptr;                             // recorded behind the plan's last launches
		~ChunkPlan();
	};
	std::vector<std::unique_ptr<ChunkPlan>> chunk_plans;   // most recently used first, at most 8, keyed on (frames, K, len)
	std::vector<std::unique_ptr<ChunkPlan>> retired_plans; // evicted, freed when their event has completed
	bool choose_chunks(long frames, int *K, long *len) const;
	CascadeParams params(const double *in, long in_stride, ssize_t frames, double *out, long out_stride) const;
	bool wire_ok(int in_fmt, bool sink_on, int out_fmt, const void *in, long in_stride, const void *out, long out_stride, ssize_t frames) const;
	ChunkPlan *chunk_plan_for(long frames, int K, long len);
	bool fuse_gtable(ChunkPlan &plan);   // the plan's G table (matrix-core prepass of the fused path), made on first use; false: not available (more than 16 sections)
	bool build_chunk_plan(ChunkPlan &chunk, long frames, int K, long len);
};

class RemixStage : public Stage {
public:
	bool init(const Spec &sp);
	const char *type() const override { return weighted ? "mix" : "remix"; }
	std::string describe() const override { return std::string(type()) + "[" + std::to_string(ch_in) + "->" + std::to_string(ch_out) + "]"; }
	ssize_t run(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st) override;
	void reset(hipStream_t) override {}
	// element-wise kernels speak every wire format at either end
	bool wire_in_ok(int, const void *, long, ssize_t, bool, int) const override { return wire_fusion_on(); }
	bool wire_out_ok(int, const void *, long, ssize_t, bool, int) const override { return wire_fusion_on(); }
	size_t device_bytes() const override { return d_idx.bytes + d_w.bytes + d_post.bytes; }
	// for the resident small-block wave: [ch_out][max_n] source channels, -1 terminated; of a weighted mix also the weights and the factors behind the sums
	const int *device_idx() const { return d_idx.as<int>(); }
	const double *device_w() const { return weighted ? d_w.as<double>() : nullptr; }
	const double *device_post() const { return (weighted && d_post.p) ? d_post.as<double>() : nullptr; }
	int sources_per_row() const { return max_n; }
private:
	DevBuf d_idx, d_w, d_post;           // d_w / d_post: weighted rows (Kind::Mix)
	int max_n = 1;
	bool weighted = false;
};

// integer per-channel delay with carried state + end-of-chain discard (the reference's `align`, align.c)
class DelayStage : public Stage {
public:
	bool init(const Spec &sp);
	const char *type() const override { return "align"; }
	std::string describe() const override;
	ssize_t run(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st) override;
	void reset(hipStream_t st) override;
	bool wire_in_ok(int, const void *, long, ssize_t, bool, int) const override { return wire_fusion_on(); }
	bool wire_out_ok(int, const void *, long, ssize_t, bool, int) const override { return wire_fusion_on(); }
	size_t device_bytes() const override { return ring.bytes; }
private:
	std::vector<ssize_t> len;
	ssize_t discard = 0, remaining_discard = 0;
	long ring_per_stream = 0, max_len = 0, pos = 0;
	int phase = 0;
	DevBuf d_len, d_off, ring;
};

// conv.cpp: transform size of the FFT convolver for a T-tap filter and calls of max_frames frames (+ relative cost)
long conv_plan(long T, long max_frames, bool resampler, double *cost);

// resample.cpp
void resample_polyphase_table(const Spec &sp, int *J, long *out_delay, std::vector<double> &tab);

// conv.cpp: FirDirect / Conv / Resample.  `prev` (may be null) is the stage immediately before: when it is another FFT
// convolver on the same channel pairs its K3 writes this stage's ring directly.  `feeder` (may be null) is the cascade stage immediately before,
// which can write straight into the convolver's planar ring instead of an interleaved slab.
Stage *make_conv_stage(const Spec &sp, int n_streams, ssize_t max_frames, CascadeStage *feeder, Stage *prev);
// a direct-form FIR stage (conv.cpp) as a pass of the resident small-block wave: taps, filter of every channel, history; *phase: which half of the history is current
bool fir_direct_view(Stage *s, struct ResidentPass *ps, const int **phase);

}  // namespace dspamd
