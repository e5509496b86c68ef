// chain.h -- stand-alone host-side chain runtime over the plugin ABI (what effects_chain.c does for the
// reference's host): chain-language parsing, merge optimisation, latency alignment, drain accounting.
#pragma once
#include <string>
#include <vector>
#include "plugin.h"

namespace dspamd {

struct ChainPlan {
	std::vector<struct effect *> effects;   // owned; prev/next linked
	stream_info istream{ 0, 0 }, ostream{ 0, 0 };
	ssize_t drain_frames = 0;               // input-rate zero frames to push at end of stream
	ssize_t zero_ref = 0;
	~ChainPlan();
	std::vector<const Spec *> specs() const;
};

// build_effects_chain_from_string() + build_effects_chain_finish()  (effects_chain.c:934-984)
bool build_chain(const char *chain_str, int fs, int channels, const char *dir, ChainPlan &plan);

}  // namespace dspamd
