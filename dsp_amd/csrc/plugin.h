// plugin.h -- the `struct effect` objects handed to a host through the reference's plugin ABI.
#pragma once
#include <memory>
#include "dsp_effect_abi.h"
#include "effects.h"
#include "engine.h"

namespace dspamd {

constexpr unsigned NODE_MAGIC = 0x44535041u;   // "DSPA"

// what e->data points to for every effect this library creates
struct Node {
	unsigned magic = NODE_MAGIC;
	SpecPtr spec;
	std::unique_ptr<Pipeline> pipe;      // single-effect, single-stream pipeline for run() on host buffers
	ssize_t pipe_frames = 0;
	DevBuf d_in, d_out;
	ssize_t out_cap_frames = 0;
	bool draining = false;
};

// wrap a Spec into a calloc'd struct effect (run == NULL when noop)
struct effect *make_effect(SpecPtr spec, bool noop);
// nullptr if e was not created by this library
Node *node_of(struct effect *e);
struct effect *make_align_effect(int fs, int channels, const std::vector<ssize_t> &len, ssize_t discard);

}  // namespace dspamd
