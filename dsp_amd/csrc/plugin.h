// plugin.h -- the `struct effect` objects handed to a host through the reference's plugin ABI.
#pragma once
#include <algorithm>
#include <atomic>
#include <memory>
#include "dsp_effect_abi.h"
#include "effects.h"
#include "engine.h"

namespace dspamd {

constexpr unsigned NODE_MAGIC = 0x44535041u;   // "DSPA"

// A device-resident segment of the host's chain: consecutive effects of this library (linked through e->prev /
// e->next when the first of them runs) compiled into ONE fused pipeline -- one H2D copy at the head, the whole
// segment on the device (cascade fusion, LTI merges, convolvers feeding each other, exactly as in the batch API),
// one D2H copy; the other members' run() hand the result through.  SURVEY.md section 7 step 4 / section 8(f).
// Small blocks of a segment made of remixes / mixes, direct FIRs and at most one cascade (the equaliser, the crossover, a crossover with correction FIRs, st2ms ...)
// through a workgroup that stays on the device for a
// few milliseconds and polls a mailbox -- no launch per block (kernels_resident.hip; the mailbox protocol: engine.h).  The wave leaves by itself (clock, loop
// bound) or when asked to; the next block starts another one.  A block it does not serve in time goes through a launch (the third such block switches the
// mechanism off for the segment).  The wave does not fence the states on the path of a block: whoever else is about to touch them calls quiesce() first.
struct Resident {
	ResidentCtl *ctl = nullptr;              // page-locked host memory: the wave's alive word
	ResidentUnit *mail_in = nullptr;         // request mailbox: device memory the CPU stores into (large BAR), else page-locked host memory
	ResidentUnit *mail_out = nullptr;        // reply mailbox: page-locked host memory
	bool in_device = false;                  // mail_in is device memory (write-combined from here: fences around the control unit)
	hipStream_t st = nullptr;
	ResidentParams rp;
	size_t lds = 0;
	unsigned seq = 0;                        // sequence number of the last request made
	long max_work = 0;                       // frames a block may have (beyond it the ordinary, time-parallel kernels are faster)
	int widest = 1;                          // channels of the widest point of the segment (a block's frames x that many samples fit a half of the wave's buffer)
	const int *fir_phase[RES_MAX_PASSES] = { nullptr, nullptr, nullptr, nullptr };     // per FIR pass: the stage's own "which half of the history is current"
	bool off = false;
	bool dirty = false;                      // the wave has served blocks since it was last waited for: the states in device memory may still be on their way
	int timeouts = 0;                        // blocks the wave did not serve in time (the third one switches the path off for the segment)
#ifdef RES_TIMING
	double t_write_us = 0.0, t_wait_us = 0.0;
#endif
	bool ready = false;                      // init() has accepted the segment; mailboxes and the stream come with the first small block (open())
	bool init(const Pipeline &pipe);         // false: the segment's stages are not all passes the wave knows
	bool takes(ssize_t frames) const { return !off && ready && frames >= 1 && (long) frames <= max_work && (size_t) frames * widest <= (size_t) rp.buf_doubles / 2 && (size_t) frames * widest <= (size_t) RESIDENT_UNITS; }
	// the block at `in` ([frames][Cin]) through the wave into `out` ([frames][C]); false: not served (the caller takes the ordinary path, on the same states)
	bool serve(const double *in, ssize_t frames, double *out);
	void stop();                             // ask the wave to leave and wait for it
	void quiesce();                          // before anything else reads or writes the cascade's states: wait until the wave says they are out (it stays)
	~Resident();
private:
	bool open();
	bool launch();
};

struct Segment {
	std::vector<struct effect *> members;    // chain order; members.front() is the head
	std::unique_ptr<Pipeline> pipe;          // single-stream pipeline for run() on host buffers
	ssize_t pipe_frames = 0, out_cap_frames = 0;
	int ch_in = 0, ch_out = 0;
	bool in_place = true;
	bool touched = false;                    // frames have gone through since the last reset
	DevBuf d_in, d_out;
	MappedPair mapped;                       // staging for small blocks (engine.h)
	std::unique_ptr<Resident> resident;      // ... and, for a segment that is one cascade, the wave that serves them without a launch
	PinnedStage staged;                      // page-locked staging for larger ones (engine.h)
	long calls = 0, small_calls = 0;         // run() calls so far / of those, blocks that a host loop would have finished sooner (the advisory below)
	bool advised = false;
	~Segment();
};

// How the plugin path served its blocks, process-wide (dspamd_plugin_counters, include/dsp_amd.h): what tests/test_gpu_resident.py asserts on
struct PluginCounters {
	std::atomic<long long> wave_blocks { 0 };        // blocks served by a resident wave (no launch)
	std::atomic<long long> mapped_blocks { 0 };      // small blocks through the mapped staging buffers and a launch
	std::atomic<long long> copied_blocks { 0 };      // larger blocks: copies (registered, staged or pageable) and launches
	std::atomic<long long> wave_launches { 0 };      // resident kernels started
	std::atomic<long long> wave_timeouts { 0 };      // blocks a wave did not serve in time (served by a launch instead)
	std::atomic<long long> wave_off { 0 };           // segments whose resident path was switched off
};
extern PluginCounters g_plugin_counters;

// what e->data points to for every effect this library creates
struct Node {
	unsigned magic = NODE_MAGIC;
	SpecPtr spec;
	std::shared_ptr<Segment> seg;            // set at the first run() of any member
};

// wrap a Spec into a calloc'd struct effect (run == NULL when noop)
struct effect *make_effect(SpecPtr spec, bool noop);
// nullptr if e was not created by this library
Node *node_of(struct effect *e);
struct effect *make_align_effect(int fs, int channels, const std::vector<ssize_t> &len, ssize_t discard);

}  // namespace dspamd
