// capi.cpp -- the stand-alone C ABI declared in include/dsp_amd.h.
#include "dsp_amd.h"
#include "chain.h"
#include "engine.h"
#include "plugin.h"
#include "pcm_params.h"
#include <cstring>
#include <cstdio>

using namespace dspamd;

struct dspamd_batch {
	ChainPlan plan;
	std::unique_ptr<Pipeline> pipe;
	std::string plan_str;
	ssize_t drain_left = 0;
	ssize_t iframes = 0;
	DevBuf zeros;
	ssize_t zeros_frames = 0;
	ssize_t wire_frames_out = 0;      // frames of each stream written through the sink so far (position in the dither sequences)
	int wire_fused = 0;
};

struct dspamd_chain {
	dspamd_batch *b = nullptr;
	DevBuf d_in, d_out;
	MappedPair mapped;
	PinnedStage staged;
	ssize_t cap = 0, out_cap = 0;
};

extern "C" {

const char *dspamd_version(void) { return "dsp_amd 0.1 (gfx950)"; }
const char *dspamd_last_error(void) { return last_error(); }
int dspamd_device_count(void) { return device_count(); }

int dspamd_set_device(int device)
{
	return hip_ok(hipSetDevice(device), "hipSetDevice") ? 0 : -1;
}

void dspamd_set_loglevel(int level) { g_loglevel = level; }

const struct effect_info *dspamd_get_effect_info(const char *name) { return registry_lookup(name); }

// ---------------------------------------------------------------- host-side planning (no device needed)

// Build the chain plan (parse, merge, prepare, alignment, drain accounting) and copy the FIR the effect at index
// `effect` ended up with for output channel `channel` (fir / fir_p / hilbert taps, or the FIR a reverse-IIR effect
// was designed into).  Returns the number of taps (which may exceed max_taps: nothing beyond max_taps is written),
// 0 when that channel is not filtered by that effect, -1 on error.  *delay (optional) = the effect's delay on that
// channel as reported to the host (channel_offsets).
ssize_t dspamd_plan_fir(const char *chain_str, int fs, int channels, const char *dir, int effect, int channel, double *taps, ssize_t max_taps, ssize_t *delay)
{
	ChainPlan plan;
	if (!build_chain(chain_str, fs, channels, dir, plan)) return -1;
	const std::vector<const Spec *> specs = plan.specs();
	if (effect < 0 || effect >= (int) specs.size()) { set_error("plan_fir: no effect %d (chain has %zu)", effect, specs.size()); return -1; }
	const Spec &sp = *specs[effect];
	if ((sp.kind != Kind::Conv && sp.kind != Kind::FirDirect) || channel < 0 || channel >= sp.ch_in) { set_error("plan_fir: effect %d (%s) is not an FIR stage", effect, sp.name.c_str()); return -1; }
	if (!sp.sel[channel]) return 0;
	int f = 0;
	if (sp.fch > 1) for (int k = 0; k < channel; ++k) if (sp.sel[k]) ++f;
	for (ssize_t i = 0; i < sp.T && i < max_taps; ++i) taps[i] = sp.taps[(size_t) i * sp.fch + f];
	if (delay) *delay = sp.ch_latency.empty() ? (sp.latency + sp.ref) : sp.ch_latency[channel];
	return sp.T;
}

// ---------------------------------------------------------------- batch

dspamd_batch *dspamd_batch_create(const char *chain_str, int fs, int channels, int n_streams, ssize_t max_frames, const char *dir)
{
	if (n_streams < 1 || max_frames < 1) { set_error("batch: invalid n_streams/max_frames"); return nullptr; }
	if (device_count() < 1) { set_error("batch: no HIP device available (the GPU backend has no CPU fallback)"); return nullptr; }
	std::unique_ptr<dspamd_batch> b(new dspamd_batch);
	if (!build_chain(chain_str, fs, channels, dir, b->plan)) return nullptr;
	b->pipe = Pipeline::compile(b->plan.specs(), fs, channels, n_streams, max_frames);
	if (!b->pipe) return nullptr;
	b->plan_str = b->pipe->plan();
	b->drain_left = b->plan.drain_frames;
	log_msg(LL_VERBOSE, "batch: %s", b->plan_str.c_str());
	return b.release();
}

int dspamd_batch_out_fs(dspamd_batch *b) { return b->pipe->fs_out; }
int dspamd_batch_out_channels(dspamd_batch *b) { return b->pipe->ch_out; }
ssize_t dspamd_batch_max_out_frames(dspamd_batch *b, ssize_t in_frames) { return b->pipe->max_out_frames(in_frames); }
ssize_t dspamd_batch_drain_frames(dspamd_batch *b) { return b->plan.drain_frames; }
const char *dspamd_batch_plan(dspamd_batch *b) { return b->plan_str.c_str(); }
int dspamd_batch_n_stages(dspamd_batch *b) { return b->pipe->n_stages(); }

ssize_t dspamd_batch_run(dspamd_batch *b, const void *d_in, ssize_t frames, void *d_out, ssize_t out_stride_frames, void *stream)
{
	if (frames < 1) return 0;
	b->iframes += frames;
	return b->pipe->run(static_cast<const double *>(d_in), frames, static_cast<double *>(d_out), (long) out_stride_frames, static_cast<hipStream_t>(stream));
}

ssize_t dspamd_batch_run_strided(dspamd_batch *b, const void *d_in, ssize_t in_stride_frames, ssize_t frames, void *d_out, ssize_t out_stride_frames, void *stream)
{
	if (frames < 1) return 0;
	b->iframes += frames;
	return b->pipe->run(static_cast<const double *>(d_in), frames, static_cast<double *>(d_out), (long) out_stride_frames, static_cast<hipStream_t>(stream), (long) in_stride_frames);
}

ssize_t dspamd_batch_drain(dspamd_batch *b, ssize_t block_frames, void *d_out, ssize_t out_stride_frames, void *stream)
{
	// drain_effects_chain(), effects_chain.c:1186-1218
	if (b->iframes < 1 || block_frames < 1) return -1;
	block_frames = std::min<ssize_t>(block_frames, b->pipe->max_frames);
	hipStream_t st = static_cast<hipStream_t>(stream);
	if (b->drain_left > 0) {
		const ssize_t n = std::min(block_frames, b->drain_left);
		if (b->zeros_frames < n) {
			// (as much silence as one drain call pushes, not a whole block of it: 1 GB instead of 16 at the bench's step)
			const ssize_t want = std::min(block_frames, b->plan.drain_frames);
			if (!b->zeros.alloc((size_t) b->pipe->S * want * b->pipe->ch_in * sizeof(double), true)) return -2;
			b->zeros_frames = want;
		}
		b->drain_left -= n;
		return b->pipe->run(b->zeros.as<double>(), n, static_cast<double *>(d_out), (long) out_stride_frames, st);
	}
	long stride = (long) out_stride_frames;
	if (stride <= 0) stride = (long) b->pipe->max_out_frames(block_frames);
	return b->pipe->drain2(block_frames, static_cast<double *>(d_out), stride, st);
}

static WireSink make_sink(dspamd_batch *b, int out_fmt, int dither_prec, void *d_stats)
{
	WireSink k;
	k.on = 1; k.fmt = out_fmt;
	// tpdf_dither_get_mult(), util.h:157-163
	k.dither_mult = (dither_prec >= 1 && dither_prec <= 32) ? 1.0 / ((double) 0x7fffffff * (double) (1u << (dither_prec - 1))) : 0.0;
	k.samples_before = (long) b->wire_frames_out * b->pipe->ch_out;
	k.stats = static_cast<double *>(d_stats);
	return k;
}

ssize_t dspamd_batch_run_wire(dspamd_batch *b, int in_fmt, const void *d_in, ssize_t in_stride_frames, ssize_t frames,
                              int out_fmt, void *d_out, ssize_t out_stride_frames, int dither_prec, void *d_stats, void *stream)
{
	if (frames < 1) return 0;
	if (!d_in) { set_error("batch_run_wire: no input"); return -1; }
	b->iframes += frames;
	const ssize_t f = b->pipe->run_wire(in_fmt, d_in, (long) in_stride_frames, frames, make_sink(b, out_fmt, dither_prec, d_stats), d_out, (long) out_stride_frames,
	                                    static_cast<hipStream_t>(stream), &b->wire_fused);
	if (f > 0) b->wire_frames_out += f;
	return f;
}

ssize_t dspamd_batch_drain_wire(dspamd_batch *b, ssize_t block_frames, int out_fmt, void *d_out, ssize_t out_stride_frames, int dither_prec, void *d_stats, void *stream)
{
	if (b->iframes < 1 || block_frames < 1) return -1;
	block_frames = std::min<ssize_t>(block_frames, b->pipe->max_frames);
	hipStream_t st = static_cast<hipStream_t>(stream);
	ssize_t f;
	if (b->drain_left > 0) {
		const ssize_t n = std::min(block_frames, b->drain_left);
		if (b->zeros_frames < n) {
			// (as much silence as one drain call pushes, not a whole block of it: 1 GB instead of 16 at the bench's step)
			const ssize_t want = std::min(block_frames, b->plan.drain_frames);
			if (!b->zeros.alloc((size_t) b->pipe->S * want * b->pipe->ch_in * sizeof(double), true)) return -2;
			b->zeros_frames = want;
		}
		b->drain_left -= n;
		f = b->pipe->run_wire(PCM_DOUBLE, b->zeros.p, 0, n, make_sink(b, out_fmt, dither_prec, d_stats), d_out, (long) out_stride_frames, st, &b->wire_fused);
	}
	else f = b->pipe->run_wire(PCM_DOUBLE, nullptr, 0, block_frames, make_sink(b, out_fmt, dither_prec, d_stats), d_out, (long) out_stride_frames, st, &b->wire_fused);
	if (f > 0) b->wire_frames_out += f;
	return f;
}

int dspamd_batch_wire_fused(dspamd_batch *b) { return b->wire_fused; }

void dspamd_batch_reset(dspamd_batch *b, void *stream)
{
	b->pipe->reset(static_cast<hipStream_t>(stream));
	b->drain_left = b->plan.drain_frames;
	b->iframes = 0;
	b->wire_frames_out = 0;
}

void dspamd_batch_destroy(dspamd_batch *b)
{
	if (!b) return;
	(void) hipDeviceSynchronize();
	delete b;
}

// ---------------------------------------------------------------- chain (host buffers, one stream)

dspamd_chain *dspamd_chain_build(const char *chain_str, int fs, int channels, const char *dir, int *out_fs, int *out_channels)
{
	const ssize_t cap = 1 << 16;
	dspamd_batch *b = dspamd_batch_create(chain_str, fs, channels, 1, cap, dir);
	if (!b) return nullptr;
	dspamd_chain *c = new dspamd_chain;
	c->b = b;
	c->cap = cap;
	c->out_cap = std::max<ssize_t>(b->pipe->max_out_frames(cap), 1);
	if (!c->d_in.alloc((size_t) cap * channels * sizeof(double), false) || !c->d_out.alloc((size_t) c->out_cap * b->pipe->ch_out * sizeof(double), false)) {
		dspamd_chain_destroy(c);
		return nullptr;
	}
	c->mapped.alloc();
	if (out_fs) *out_fs = b->pipe->fs_out;
	if (out_channels) *out_channels = b->pipe->ch_out;
	return c;
}

ssize_t dspamd_chain_run(dspamd_chain *c, const double *in, ssize_t frames, double *out, ssize_t out_capacity_frames)
{
	const int ci = c->b->pipe->ch_in, co = c->b->pipe->ch_out;
	ssize_t done = 0, produced = 0;
	if (frames > 0 && frames <= c->cap && c->mapped.fits((size_t) frames * ci * sizeof(double), (size_t) c->b->pipe->max_out_frames(frames) * co * sizeof(double))) {
		// small block: no copy commands (engine.h, MappedPair)
		memcpy(c->mapped.in, in, (size_t) frames * ci * sizeof(double));
		const ssize_t f = dspamd_batch_run(c->b, c->mapped.in, frames, c->mapped.out, (ssize_t) (c->mapped.bytes / (co * sizeof(double))), nullptr);
		if (!c->mapped.wait_block(nullptr) || f < 0) return -1;
		if (f > out_capacity_frames) { set_error("chain_run: output capacity exceeded"); return -1; }
		if (f > 0) memcpy(out, c->mapped.out, (size_t) f * co * sizeof(double));
		return f;
	}
	if (frames > c->cap && c->staged.ensure((size_t) std::min(frames, c->cap) * ci * sizeof(double), (size_t) c->b->pipe->max_out_frames(std::min(frames, c->cap)) * co * sizeof(double))) {
		// larger blocks: through page-locked staging buffers, the calling thread copying beside the GPU's work (engine.h, PinnedStage)
		const ssize_t f = c->staged.run(in, frames, c->cap, ci, out, out_capacity_frames, co, c->d_in.p, c->d_out.p, nullptr,
		                                [&](const double *di, ssize_t nb, double *dout) { return dspamd_batch_run(c->b, di, nb, dout, c->out_cap, nullptr); });
		if (f == -2) { set_error("chain_run: output capacity exceeded"); return -1; }
		return f;
	}
	while (done < frames) {
		const ssize_t nb = std::min(frames - done, c->cap);
		if (!hip_ok(hipMemcpy(c->d_in.p, in + done * ci, (size_t) nb * ci * sizeof(double), hipMemcpyHostToDevice), "H2D")) return -1;
		const ssize_t f = dspamd_batch_run(c->b, c->d_in.p, nb, c->d_out.p, c->out_cap, nullptr);
		if (f < 0) return f;
		if (produced + f > out_capacity_frames) { set_error("chain_run: output capacity exceeded"); return -1; }
		if (f > 0 && !hip_ok(hipMemcpy(out + produced * co, c->d_out.p, (size_t) f * co * sizeof(double), hipMemcpyDeviceToHost), "D2H")) return -1;
		produced += f;
		done += nb;
	}
	return produced;
}

ssize_t dspamd_chain_drain(dspamd_chain *c, ssize_t block_frames, double *out, ssize_t out_capacity_frames)
{
	const int co = c->b->pipe->ch_out;
	const ssize_t f = dspamd_batch_drain(c->b, std::min(block_frames, c->cap), c->d_out.p, c->out_cap, nullptr);
	if (f < 0) return -1;
	if (f > out_capacity_frames) { set_error("chain_drain: output capacity exceeded"); return -2; }
	if (f > 0 && !hip_ok(hipMemcpy(out, c->d_out.p, (size_t) f * co * sizeof(double), hipMemcpyDeviceToHost), "D2H")) return -2;
	return f;
}

ssize_t dspamd_chain_max_out_frames(dspamd_chain *c, ssize_t in_frames) { return c->b->pipe->max_out_frames(in_frames); }
ssize_t dspamd_chain_drain_frames(dspamd_chain *c) { return c->b->plan.drain_frames; }

void dspamd_chain_reset(dspamd_chain *c)
{
	dspamd_batch_reset(c->b, nullptr);
	(void) hipStreamSynchronize(nullptr);
}

void dspamd_chain_destroy(dspamd_chain *c)
{
	if (!c) return;
	dspamd_batch_destroy(c->b);
	delete c;
}

int dspamd_chain_n_effects(dspamd_chain *c) { return (int) c->b->plan.effects.size(); }

const char *dspamd_chain_effect_name(dspamd_chain *c, int i)
{
	if (i < 0 || i >= (int) c->b->plan.effects.size()) return nullptr;
	return c->b->plan.effects[i]->name;
}

// ---------------------------------------------------------------- how the plugin path served its blocks

int dspamd_plugin_counters(long long *out, int n)
{
	const PluginCounters &c = g_plugin_counters;
	const long long v[6] = { c.wave_blocks.load(), c.mapped_blocks.load(), c.copied_blocks.load(), c.wave_launches.load(), c.wave_timeouts.load(), c.wave_off.load() };
	int k = 0;
	for (; k < n && k < 6; ++k) out[k] = v[k];
	return k;
}

// ---------------------------------------------------------------- per-kernel HIP-event timing

void dspamd_profile_enable(int on) { g_prof.on = (on != 0); }

// Synchronises the device, then returns a JSON-ish text: one "name total_ms launches" line per kernel seen since
// the last call.  The pointer stays valid until the next call.
const char *dspamd_profile_collect(void)
{
	static std::string text;
	(void) hipDeviceSynchronize();
	std::vector<std::string> names;
	std::vector<double> ms;
	std::vector<long> counts;
	g_prof.collect(names, ms, counts);
	text.clear();
	char line[256];
	for (size_t i = 0; i < names.size(); ++i) {
		snprintf(line, sizeof(line), "%s %.6f %ld\n", names[i].c_str(), ms[i], counts[i]);
		text += line;
	}
	return text.c_str();
}

// ---------------------------------------------------------------- bench endpoints

int dspamd_sgen_sweep(void *d_buf, int n_streams, ssize_t frames, int channels, int fs, double freq0, double freq1, double dfreq, ssize_t total_frames, ssize_t pos0, void *stream)
{
	if (!(freq0 > 0.0) || !(freq1 > 0.0)) { set_error("sgen: frequencies must be > 0"); return -1; }
	launch_sgen(static_cast<double *>(d_buf), n_streams, frames, channels, fs, 1, freq0, freq1, dfreq, total_frames, 0, 0, pos0, static_cast<hipStream_t>(stream));
	return hip_ok(hipGetLastError(), "sgen") ? 0 : -1;
}

int dspamd_sgen_delta(void *d_buf, int n_streams, ssize_t frames, int channels, ssize_t offset, ssize_t doffset, ssize_t pos0, void *stream)
{
	launch_sgen(static_cast<double *>(d_buf), n_streams, frames, channels, 48000, 2, 1.0, 1.0, 0.0, 0, offset, doffset, pos0, static_cast<hipStream_t>(stream));
	return hip_ok(hipGetLastError(), "sgen") ? 0 : -1;
}

int dspamd_sgen_sine(void *d_buf, int n_streams, ssize_t frames, int channels, int fs, double freq0, double dfreq, ssize_t pos0, void *stream)
{
	launch_sgen_sine(static_cast<double *>(d_buf), n_streams, frames, channels, fs, freq0, dfreq, pos0, static_cast<hipStream_t>(stream));
	return hip_ok(hipGetLastError(), "sgen") ? 0 : -1;
}

int dspamd_digest(const void *d_buf, int n_streams, ssize_t frames, ssize_t stride_frames, int channels, void *d_out, void *stream)
{
	launch_digest(static_cast<const double *>(d_buf), n_streams, frames, stride_frames, channels, static_cast<double *>(d_out), static_cast<hipStream_t>(stream));
	return hip_ok(hipGetLastError(), "digest") ? 0 : -1;
}

// ---------------------------------------------------------------- wire formats on the device

static size_t pcm_bytes(int fmt) { return pcm_sample_bytes(fmt); }

size_t dspamd_pcm_sample_bytes(int fmt) { return pcm_bytes(fmt); }

int dspamd_pcm_read(int fmt, const void *d_in, void *d_out, ssize_t n_samples, void *stream)
{
	if (!pcm_bytes(fmt)) { set_error("pcm_read: unknown format %d", fmt); return -1; }
	if (device_count() < 1) { set_error("pcm_read: no HIP device available"); return -1; }
	PcmReadParams p{ d_in, static_cast<double *>(d_out), (long) n_samples, (long) n_samples, (long) n_samples, 1, fmt };
	launch_pcm_read(p, 1, static_cast<hipStream_t>(stream));
	return hip_ok(hipGetLastError(), "pcm_read") ? 0 : -1;
}

int dspamd_pcm_write(int fmt, const void *d_in, ssize_t in_stride_frames, void *d_out, int n_streams, ssize_t frames, int channels,
                     int dither_prec, ssize_t frames_before, void *d_stats, void *stream)
{
	if (!pcm_bytes(fmt)) { set_error("pcm_write: unknown format %d", fmt); return -1; }
	if (device_count() < 1) { set_error("pcm_write: no HIP device available"); return -1; }
	PcmWriteParams p;
	p.in = static_cast<const double *>(d_in);
	p.out = d_out;
	p.in_stride_frames = in_stride_frames; p.out_stride_frames = frames; p.frames = frames;
	p.C = channels; p.fmt = fmt;
	// tpdf_dither_get_mult(), util.h:157-163
	p.dither_mult = (dither_prec >= 1 && dither_prec <= 32) ? 1.0 / ((double) 0x7fffffff * (double) (1u << (dither_prec - 1))) : 0.0;
	p.samples_before = (long) frames_before * channels;
	p.stats = static_cast<double *>(d_stats);
	launch_pcm_write(p, n_streams, static_cast<hipStream_t>(stream));
	return hip_ok(hipGetLastError(), "pcm_write") ? 0 : -1;
}

int dspamd_copy_probe(const void *d_src, void *d_dst, size_t bytes, void *stream)
{
	launch_copy_probe(d_src, d_dst, bytes, static_cast<hipStream_t>(stream));
	return hip_ok(hipGetLastError(), "copy_probe") ? 0 : -1;
}

}  // extern "C"
