// engine.h -- the device pipeline: a chain of Specs compiled into fused GPU stages that process a
// batch of S independent streams whose buffers ([stream][frame][channel] fp64) live in HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <memory>
#include <string>
#include <vector>
#include <cstring>
#include <algorithm>
#include "effects.h"
#include "kparams.h"

namespace dspamd {

bool hip_ok(hipError_t e, const char *what);   // logs + set_error on failure
int device_count();

void trace_mem(const char *what, const void *p, size_t n);      // DSP_AMD_TRACE_MEM (engine.cpp)

struct DevBuf {
	void *p = nullptr;
	size_t bytes = 0;
	bool canary = false;                        // DSP_AMD_GUARD=3: a page of a pattern on both sides, checked when the buffer is released
	const void *site = nullptr;
	DevBuf() = default;
	DevBuf(const DevBuf &) = delete;
	DevBuf &operator=(const DevBuf &) = delete;
	~DevBuf() { release(); }
	bool alloc(size_t n, bool zero = true);
	bool upload(const void *src, size_t n);    // alloc + H2D (synchronous)
	void release();
	template <class T> T *as() const { return static_cast<T *>(p); }
};

// Small host blocks (LADSPA hosts: 64 ... 1024 frames): two page-locked, device-mapped staging buffers that the first kernel
// reads and the last kernel writes over PCIe directly -- no copy commands, one launch sequence and one wait per call.  Measured
// (docs/history.md section 6): faster than H2D / D2H copy commands up to a few tens of KB per block, slower at 128 KB; the limit is
// DSP_AMD_PLUGIN_MAPPED_KB (default 32, 0 = off).
struct MappedPair {
	double *in = nullptr, *out = nullptr;
	size_t bytes = 0;                           // of each; 0 = not available
	MappedPair() = default;
	MappedPair(const MappedPair &) = delete;
	MappedPair &operator=(const MappedPair &) = delete;
	~MappedPair();
	void alloc();                               // failure just leaves bytes == 0
	bool fits(size_t in_bytes, size_t out_bytes) const { return bytes && in_bytes <= bytes && out_bytes <= bytes; }
	// end of a small block without the runtime's stream wait: a stream memory operation writes the block's sequence number into a
	// mapped word behind the kernels, the host thread spins on it (falls back to hipStreamSynchronize when the word does not
	// arrive within a few milliseconds or the operation is not available)
	volatile unsigned *flag = nullptr;
	unsigned seq = 0;
	bool flag_off = false;
	bool wait_block(hipStream_t st);
};

// Host blocks of MORE than one pipeline call (the stand-alone host API takes buffers of any length; a host that hands over more than
// the segment was planned for): page-locked staging buffers of this library, two each way -- the calling thread fills chunk k + 1 and
// empties chunk k - 1 while the GPU copies and works on chunk k (8 ch x 2^20 frames: 168 -> 469 Msamples/s; plain copy commands on
// pageable memory serialise copy and work).  Chunks are the pipeline's own call size (the plans of the stages are made for it).
// A block of one chunk keeps the copy commands: alone, the thread's two copies only add to the call.  DSP_AMD_PLUGIN_STAGE=0 or a
// chunk beyond 64 MB: copy commands throughout.
// memcpy of a staged block on several cores: one thread moves 7 GB/s each way, the link 50 -- blocks of 1 MB and more are cut into
// slices for three helper threads (started on first use, asleep between bursts) beside the caller.  Plain memcpy when the process
// may run on fewer than four cores, when another thread is using the helpers, or for short blocks.
void crew_memcpy(void *dst, const void *src, size_t bytes);

struct PinnedStage {
	char *in[2] = { nullptr, nullptr }, *out[2] = { nullptr, nullptr };
	size_t in_cap = 0, out_cap = 0;
	hipEvent_t done[2] = { nullptr, nullptr };
	hipEvent_t copied[2] = { nullptr, nullptr };    // recorded right behind the host-to-device copy out of in[i]: the buffer may be refilled once it has passed
	bool off = false;
	PinnedStage() = default;
	PinnedStage(const PinnedStage &) = delete;
	PinnedStage &operator=(const PinnedStage &) = delete;
	~PinnedStage();
	bool ensure(size_t in_bytes, size_t out_bytes);     // buffers for chunks of this size (false: not available, use copy commands)
	// run `frames` frames of `in` through run_chunk(device in, frames, device out) -> frames produced (< 0: failed) in chunks of at most
	// `chunk` frames; d_in / d_out are the caller's device buffers of a chunk.  Returns the frames produced or -1.
	template <class F>
	ssize_t run(const double *in, ssize_t frames, ssize_t chunk, int ch_in, double *out, ssize_t out_capacity_frames, int ch_out,
	            void *d_in, void *d_out, hipStream_t st, F run_chunk);
};

template <class F>
ssize_t PinnedStage::run(const double *in, ssize_t frames, ssize_t chunk, int ch_in, double *out, ssize_t out_capacity_frames, int ch_out,
                         void *d_in, void *d_out, hipStream_t st, F run_chunk)
{
	const size_t fi = (size_t) ch_in * sizeof(double), fo = (size_t) ch_out * sizeof(double);
	ssize_t produced = 0, prev_f = 0;
	int prev = -1;
	auto collect = [&]() -> bool {                  // the results of the chunk before: wait for its copy, hand them over
		if (prev < 0) return true;
		if (!hip_ok(hipEventSynchronize(done[prev]), "staged D2H wait")) return false;
		if (prev_f > 0) crew_memcpy(out + (size_t) produced * ch_out, this->out[prev], (size_t) prev_f * fo);
		produced += prev_f;
		prev = -1;
		return true;
	};
	ssize_t done_frames = 0;
	int k = 0;
	bool in_flight[2] = { false, false };               // in[i] has a copy command queued whose `copied` event has not been waited for
	if (frames > 0) crew_memcpy(this->in[0], in, (size_t) std::min(frames, chunk) * fi);
	while (done_frames < frames) {
		const ssize_t nb = std::min(frames - done_frames, chunk);
		const int b = k & 1;
		if (!hip_ok(hipMemcpyAsync(d_in, this->in[b], (size_t) nb * fi, hipMemcpyHostToDevice, st), "staged H2D") || !hip_ok(hipEventRecord(copied[b], st), "staged H2D event")) return -1;
		in_flight[b] = true;
		const ssize_t f = run_chunk(static_cast<const double *>(d_in), nb, static_cast<double *>(d_out));
		if (f < 0) { (void) hipStreamSynchronize(st); return -1; }
		if (produced + prev_f * (prev >= 0 ? 1 : 0) + f > out_capacity_frames) { (void) hipStreamSynchronize(st); return -2; }
		if (f > 0 && !hip_ok(hipMemcpyAsync(this->out[b], d_out, (size_t) f * fo, hipMemcpyDeviceToHost, st), "staged D2H")) return -1;
		if (!hip_ok(hipEventRecord(done[b], st), "staged D2H event")) return -1;
		done_frames += nb;
		// while the GPU works on this chunk: the next one in -- once the copy command that last read that buffer (chunk k - 1's) has
		// passed: its own event, not an assumption about how far the stream has got -- and the one before out
		if (done_frames < frames) {
			if (in_flight[b ^ 1]) { if (!hip_ok(hipEventSynchronize(copied[b ^ 1]), "staged H2D wait")) return -1; in_flight[b ^ 1] = false; }
			crew_memcpy(this->in[b ^ 1], in + (size_t) done_frames * ch_in, (size_t) std::min(frames - done_frames, chunk) * fi);
		}
		if (!collect()) return -1;
		prev = b; prev_f = f;
		++k;
	}
	return collect() ? produced : -1;
}

// Optional per-kernel timing with HIP events recorded on the SAME stream the kernels are launched on
// (bench.py's roofline object).  Off by default; when on, every launch site brackets itself.
struct Profiler {
	struct Rec { const char *name; hipEvent_t a, b; };
	bool on = false;
	std::vector<Rec> recs;
	void begin(const char *name, hipStream_t st);
	void end(hipStream_t st);
	void rename_last(const char *name) { if (!recs.empty()) recs.back().name = name; }
	// after a device sync: accumulate (name -> total ms, count), then drop the events
	void collect(std::vector<std::string> &names, std::vector<double> &ms, std::vector<long> &counts);
	~Profiler();
};
extern Profiler g_prof;
struct ProfScope {
	hipStream_t st; bool on;
	ProfScope(const char *name, hipStream_t s) : st(s), on(g_prof.on) { if (on) g_prof.begin(name, st); }
	~ProfScope() { if (on) g_prof.end(st); }
	void rename(const char *name) { if (on && name) g_prof.rename_last(name); }   // the launcher knows which kernel it picked
};

// what the run / drain entry points return besides a frame count: "dry" is the reference's end-of-drain sentinel
// (effects_chain.c:1186-1218: *frames = -1), a failed launch or allocation must never look like it
enum : ssize_t { PIPE_DRY = -1, PIPE_FAILED = -2 };

// One fused device stage.  in/out are [S][stride][C] slabs; a stage may be run in place when
// in_place_ok() (out == in, same stride).
class Stage {
public:
	int S = 1, fs_in = 0, fs_out = 0, ch_in = 0, ch_out = 0;
	virtual ~Stage() {}
	virtual const char *type() const = 0;
	virtual std::string describe() const = 0;
	virtual bool in_place_ok() const { return false; }
	virtual ssize_t max_out_frames(ssize_t in_frames) const { return in_frames; }
	// returns frames produced per stream (>= 0) or PIPE_FAILED
	virtual ssize_t run(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st) = 0;
	// second-phase drain for stages that hold frames back (resample drain2); PIPE_DRY = nothing (left), PIPE_FAILED = error
	virtual ssize_t drain2(ssize_t max_frames, double *out, long out_stride, hipStream_t st) { (void) max_frames; (void) out; (void) out_stride; (void) st; return -1; }
	virtual void reset(hipStream_t st) = 0;
	virtual size_t device_bytes() const { return 0; }
	// an end-of-chain alignment that only drops the first d frames of the stream (align.c:53-62 with no channel delayed: what is
	// left of a `fir` / zita_convolver latency) can be taken over by the stage in front of it: true = done, no pass of its own
	virtual bool absorb_discard(long d) { (void) d; return false; }
	// Wire formats at the ends of a pipeline (Pipeline::run_wire): a first stage that answers wire_in_ok() reads samples of
	// `wire_in_fmt` through its `in` pointer for this one call, a last stage that answers wire_out_ok() applies `wire_sink` (dither,
	// clip, conversion: kparams.h) in its stores and writes samples of that format through `out`.  Both answers are pure
	// functions of the call; the strides stay in frames.  The pipeline sets the two members around run() and clears them after.
	virtual bool wire_in_ok(int fmt, const void *in, long in_stride, ssize_t frames, bool also_out, int out_fmt) const { (void) fmt; (void) in; (void) in_stride; (void) frames; (void) also_out; (void) out_fmt; return false; }
	virtual bool wire_out_ok(int fmt, const void *out, long out_stride, ssize_t frames, bool also_in, int in_fmt) const { (void) fmt; (void) out; (void) out_stride; (void) frames; (void) also_in; (void) in_fmt; return false; }
	int wire_in_fmt = PCM_DOUBLE;
	WireSink wire_sink = { 0, PCM_DOUBLE, 0.0, 0, nullptr };
};

class ConvStage;

class Pipeline {
public:
	// specs are borrowed for the duration of the call only
	static std::unique_ptr<Pipeline> compile(const std::vector<const Spec *> &specs, int fs, int channels, int n_streams, ssize_t max_frames);
	~Pipeline();
	int S, fs_in, ch_in, fs_out, ch_out;
	ssize_t max_frames;
	ssize_t max_out_frames(ssize_t in_frames) const;
	// in_stride: frames between the slabs of two streams in d_in (0 = frames: contiguous)
	ssize_t run(const double *d_in, ssize_t frames, double *d_out, long out_stride, hipStream_t st, long in_stride = 0);
	ssize_t drain2(ssize_t block_frames, double *d_out, long out_stride, hipStream_t st);   // rate-changer flush
	// The same from wire format to wire format (the file -> file path of dsp.c: read_buf_<fmt>, the chain, then dither / clip /
	// write_buf_<fmt>): d_in holds [S][in_stride][C_in] samples of in_fmt, d_out receives [S][out_stride][C_out] samples of
	// sink.fmt (sink.samples_before = samples of each stream written by earlier calls).  The first / last stage does the
	// conversion in its own loads / stores where it can (cascade_rows, K3 of a plain convolution); otherwise the stand-alone
	// conversion kernels run before / after on buffers of this pipeline.  Same samples either way, bit for bit.
	// d_in == nullptr: drain2() of the rate changers into the sink.  *fused (optional): bit 0 = input fused, bit 1 = output.
	ssize_t run_wire(int in_fmt, const void *d_in, long in_stride, ssize_t frames, const WireSink &sink, void *d_out, long out_stride, hipStream_t st, int *fused = nullptr);
	void reset(hipStream_t st);
	std::string plan() const;
	int n_stages() const { return (int) stages.size(); }
	// what a resident small-block wave (kernels_resident.hip, plugin.cpp) would do with a block of this single-stream pipeline: its stages as passes over the
	// block in LDS -- plain or weighted remixes, at most one cascade, direct FIRs -- or false when a stage is of another kind.  fir_phase[k]: where the host
	// finds, per request, which half of pass k's FIR history is current (the stage's own member), or nullptr
	bool resident_plan(struct ResidentParams *rp, const int *fir_phase[]) const;
	size_t device_bytes() const;
private:
	Pipeline() {}
	std::vector<std::unique_ptr<Stage>> stages;
	DevBuf tmp[2];
	DevBuf wire_tmp_in, wire_tmp_out;   // fp64 slabs either side when a wire format cannot be fused (allocated on first use)
	long tmp_stride[2] = { 0, 0 };   // frames
	int tmp_ch = 0;
	int drain_stage = 0;
};

// ---- small plugin blocks through a wave that stays on the device for a bounded time (kernels_resident.hip, plugin.cpp) ----
// The mailbox protocol (round 6).  Both directions are arrays of 16-byte units { value, word }: word = request ^ bits(value), request = (sequence number << 32) |
// frames -- a unit proves by itself that it belongs to the current request (a stale value under a new word does not decode to the request, and the other way
// round), so neither side needs a second, dependent trip behind a doorbell and neither side waits for its stores to be acknowledged:
//   host -> wave: mail_in[0] = the control unit { 0.0, request }, mail_in[1 + e] = sample e of the block.  In DEVICE memory where the CPU can store into it (large
//                 BAR: posted writes over PCIe, the wave polls and reads local memory -- scripts/ubench/barprobe.hip: 3.2 us against 5.6 for the round trip
//                 with a 1 KB block), else in page-locked host memory;
//   wave -> host: mail_out[e] = output sample e, in page-locked host memory (posted writes again); the host reads the units until every one decodes.
constexpr unsigned RESIDENT_STOP = 0xffffffffu;      // `frames` of a request that tells the wave to leave
constexpr int RESIDENT_UNITS = 4096;                 // payload units per direction (64 KB): frames x channels of a block the wave takes
struct ResidentUnit { double v; unsigned long long w; };
struct ResidentCtl {                                 // page-locked, device-mapped host memory
	unsigned alive;                                  // set by the host before a launch, cleared by the wave as its last store
	unsigned settled;                                // sequence number of the last block whose states (and output) are known to be out: written by the wave when it
	                                                 // finds nothing to do, behind a wait for its stores -- never on the path of a block (Resident::quiesce)
	unsigned pad[14];
};
// What the wave does with a block is a short list of passes over it in LDS (round 6; round 5: one cascade, or a plain remix in front of one):
//   remix / mix   every output channel a sum of input channels (remix.c:39-101), or a weighted one (st2ms.c:28-54, crossfeed.c:41-46), bit-exact
//   cascade       gains / adds / sections as a systolic array over the lanes of a row (at most two such passes per segment: their ops live in registers)
//   direct FIR    filters of up to 32 taps in the reference's own summation order (fir.c:43-62, fir_p.c:131-148), bit-exact; history in device memory
enum : int { RES_PASS_REMIX = 1, RES_PASS_CASCADE = 2, RES_PASS_FIR = 3 };
constexpr int RES_MAX_PASSES = 4, RES_MAX_CASCADES = 2, RES_FIR_MAX_CH = 16, RES_FIR_TAPS = 32;
struct ResidentPass {
	int kind, c_in, c_out;
	int casc;                                        // cascade: which of the segment's cascades (ResidentParams::cs)
	const int *idx;                                  // remix: [c_out][max_n] source channels, -1 terminated
	const double *w, *post;                          // a weighted mix: [c_out][max_n] weights, [c_out] factors applied to the sums (or nullptr)
	int max_n;
	int T;                                           // direct FIR: taps
	const double *taps;                              // [n_filters][RES_FIR_TAPS]
	const int *foc;                                  // [c] filter of the channel, -1 = the channel passes through
	double *hist;                                    // [2][c][RES_FIR_TAPS]: x[-(q + 1)] of the block's first frame; which half holds it comes with the request
};
// a request's low word: frames in bits 0 ... 15, bit 16 + k = the half of pass k's FIR history that is current (the ordinary kernel alternates them)
constexpr unsigned RES_FRAMES_MASK = 0xffffu;
struct ResidentParams {
	ResidentCtl *ctl;
	const ResidentUnit *mail_in;                     // [1 + RESIDENT_UNITS]
	ResidentUnit *mail_out;                          // [RESIDENT_UNITS]
	int Cin, Cout;                                   // channels of the block as it comes and as it goes
	int n_pass;
	ResidentPass pass[RES_MAX_PASSES];
	int n_casc;
	struct Casc { int C, n_ops; const OpDesc *ops; double *state; } cs[RES_MAX_CASCADES];     // channels, ops per channel, [C][n_ops] ops, [C][n_ops][2] states
	unsigned long long lifetime_ticks;               // of the 100 MHz wall clock, without a block
	unsigned long long max_life_ticks;               // ... since the launch, blocks or not (the wave leaves between two blocks; the host starts another)
	unsigned max_polls, done0;                       // hard bound on the polling loop; the sequence number already served
	int buf_doubles;                                 // doubles of the block buffer in LDS: two halves, a pass that cannot work in place goes from one to the other
	int spec_units;                                  // payload units per lane asked for together with the control unit (the blocks the host expects to send)
};
// LDS of the wave: the block buffer, two control words, a word per lane (the stores of lanes that are not a channel's last op), the FIR histories ...
// ... and a copy of every pass's tables (8 KB per pass: a FIR's taps and channel map, a remix's sources, weights and factors -- read per tap / per source,
// they must not be a trip to memory each time)
constexpr int RES_TAB_DOUBLES = 1024;
constexpr size_t resident_lds_bytes(int buf_doubles) { return ((size_t) buf_doubles + 2 + 1024 + (size_t) RES_MAX_PASSES * (RES_FIR_MAX_CH * RES_FIR_TAPS + RES_TAB_DOUBLES)) * sizeof(double); }
bool launch_cascade_resident(const ResidentParams &p, size_t lds_bytes, hipStream_t st);

// kernel launchers (kernels_*.hip)
size_t cascade_lds_bytes(int Cg, int n_ops);
const char *launch_cascade(const CascadeParams &p, int n_streams, hipStream_t stream);   // returns the name of the kernel that took the block
bool cascade_rows_takes(const CascadeParams &p, int n_streams);   // cascade_rows (the kernel that speaks the wire formats) would take this call
void launch_chunk_carry(const ChunkParams &p, int n_streams, hipStream_t stream);
void launch_chunk_fix(const ChunkParams &p, int n_streams, hipStream_t stream);
void launch_remix(const RemixParams &p, int n_streams, hipStream_t stream);
void launch_delay_ex(const DelayParams &p, long ring_alt_off, long skip, long max_len, int n_streams, hipStream_t stream);
void launch_copy_slab(const double *in, long in_stride, double *out, long out_stride, long frames, long skip, int C, int n_streams, hipStream_t stream);
void launch_sgen_sine(double *buf, int n_streams, long frames, int channels, int fs, double freq0, double dfreq, long pos0, hipStream_t stream);
void launch_sgen(double *buf, int n_streams, long frames, int channels, int fs, int kind, double freq0, double freq1, double dfreq, long total_frames, long offset, long doffset, long pos0, hipStream_t stream);
void launch_digest(const double *buf, int n_streams, long frames, long stride, int channels, double *out, hipStream_t stream);
void launch_copy_probe(const void *src, void *dst, size_t bytes, hipStream_t stream);

}  // namespace dspamd
