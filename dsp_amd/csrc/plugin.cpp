// plugin.cpp -- the reference's plugin surface (effect.h:24-59) implemented over the device pipeline.
//
// Every *_effect_init() below has the reference's name and signature (include/dsp_effect_abi.h), parses its
// argv exactly like the reference effect it replaces, and returns a calloc'd `struct effect` whose callbacks
// drive HIP kernels.  run() works on the host's interleaved fp64 buffers (one PCIe round trip per call: this
// is the compatibility path; throughput work goes through the device-resident batch API, capi.cpp).
#include "plugin.h"
#include "stages.h"
#include <ctime>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace dspamd {

static std::once_flag g_dev_once;
PluginCounters g_plugin_counters;

static void select_device_once()
{
	std::call_once(g_dev_once, [] {
		const char *d = getenv("DSP_AMD_DEVICE");
		if (d) (void) hipSetDevice(atoi(d));
	});
}

Node *node_of(struct effect *e)
{
	if (!e || !e->data) return nullptr;
	Node *n = static_cast<Node *>(e->data);
	return (n->magic == NODE_MAGIC) ? n : nullptr;
}

static void plugin_destroy(struct effect *e);

static bool is_ours(struct effect *e) { return e && e->destroy == plugin_destroy && node_of(e); }

// the segment `e` belongs to, built on first use from the chain links the host maintains
static bool ensure_segment(struct effect *e, Node *n)
{
	if (n->seg) return n->seg->pipe != nullptr;
	select_device_once();
	if (device_count() < 1) {
		set_error("%s: error: no HIP device available (the GPU backend has no CPU fallback)", e->name);
		return false;
	}
	static const bool fuse = !getenv("DSP_AMD_PLUGIN_NO_FUSE");
	struct effect *first = e, *last = e;
	if (fuse) {
		while (is_ours(first->prev) && !node_of(first->prev)->seg && first->prev->ostream.fs == first->istream.fs
		       && first->prev->ostream.channels == first->istream.channels) first = first->prev;
		while (is_ours(last->next) && !node_of(last->next)->seg && last->ostream.fs == last->next->istream.fs
		       && last->ostream.channels == last->next->istream.channels) last = last->next;
		// an integer `delay` has a no-op run() (delay.c:201-202): it may sit inside a segment but cannot head one
		while (first != e && node_of(first)->spec->kind == Kind::Delay) first = first->next;
	}
	std::shared_ptr<Segment> seg(new Segment);
	std::vector<const Spec *> specs;
	for (struct effect *m = first;; m = m->next) {
		Node *mn = node_of(m);
		seg->members.push_back(m);
		specs.push_back(mn->spec.get());
		if (mn->spec->kind == Kind::Remix || mn->spec->kind == Kind::Resample) seg->in_place = false;
		if (m == last) break;
	}
	// every member knows its segment BEFORE the first step that can fail: a failed segment (pipe == nullptr) is final --
	// no rebuild, no second FIR design and no second error line per block on what may be a real-time thread
	for (struct effect *m : seg->members) node_of(m)->seg = seg;
	for (struct effect *m : seg->members) {
		Node *mn = node_of(m);
		if (mn->spec->riir_pending && !riir_prepare(*mn->spec)) return false;    // a host that skipped prepare()
	}
	const Spec &s0 = *specs.front(), &s1 = *specs.back();
	const ssize_t cap = 1 << 16;
	seg->ch_in = s0.ch_in; seg->ch_out = s1.ch_out;
	seg->pipe = Pipeline::compile(specs, s0.fs_in, s0.ch_in, 1, cap);
	if (!seg->pipe) return false;
	seg->pipe_frames = cap;
	seg->out_cap_frames = seg->pipe->max_out_frames(cap);
	if (!seg->d_in.alloc((size_t) cap * seg->ch_in * sizeof(double), false) ||
	    !seg->d_out.alloc((size_t) seg->out_cap_frames * seg->ch_out * sizeof(double), false)) { seg->pipe.reset(); return false; }
	seg->mapped.alloc();
	{
		static const bool resident_on = [] { const char *v = getenv("DSP_AMD_PLUGIN_RESIDENT"); return !v || atoi(v) != 0; }();
		if (resident_on) {
			seg->resident.reset(new Resident);
			if (!seg->resident->init(*seg->pipe)) seg->resident.reset();
		}
	}
	if (seg->members.size() > 1) log_msg(LL_VERBOSE, "%s: info: %zu effects fused into one device segment: %s", e->name, seg->members.size(), seg->pipe->plan().c_str());
	return true;
}

// Host buffers are never registered with the HIP runtime (round 6).  Rounds 4 and 5 page-locked a block buffer the host kept handing over
// (hipHostRegister after four sightings: DMA instead of the runtime's pageable-memory staging, 10 % at the reference's 2048-frame blocks).  That is
// memory this library does not own -- the host may free it, the C library may trim or remap the heap under it -- and with the registration in the
// process a LATER, unrelated pageable copy of the same process (a torch .cpu(), an upload from a std::vector) faulted with hipErrorIllegalAddress in 3 of
// 13 one-process runs of the GPU suite; without it, in 0 of 32 (profiles/r06_ab_one_process_suite.txt, r06_one_process_runs.txt).
// Blocks that do not fit the mapped staging buffers take pageable copies (one pipeline call) or this library's own page-locked double buffers (longer ones).
Segment::~Segment() { resident.reset(); }

// ---- the resident small-block wave (see plugin.h) ----
static inline double res_now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }

bool Resident::init(const Pipeline &pipe)
{
	// the segment's stages as passes of the wave (remixes, at most one cascade, direct FIRs: Pipeline::resident_plan), or no wave for this segment
	if (!pipe.resident_plan(&rp, fir_phase)) return false;
	// (the mailboxes and the stream of its own are made by the first block the wave takes -- open(): a segment driven with larger blocks never has them)
	rp.lifetime_ticks = 300000ull;           // 3 ms of the 100 MHz clock: more than two periods of a 64-frame block at 48 kHz
	rp.max_life_ticks = 2000000ull;          // 20 ms in all: what a hipDeviceSynchronize() on another thread waits at the very most while this segment plays
	rp.max_polls = 1u << 18;                 // (a turn of the loop is a trip to the mailbox: about a second at the very most)
	rp.buf_doubles = 8192;                   // 64 KB: two halves of 4096 samples (a pass that cannot work in place goes from one to the other)
	lds = resident_lds_bytes(rp.buf_doubles);
	widest = std::max(rp.Cin, rp.Cout);
	for (int k = 0; k < rp.n_pass; ++k) widest = std::max(widest, std::max(rp.pass[k].c_in, rp.pass[k].c_out));
	// a block is frames + n_ops - 1 steps of a systolic array over the ops of a channel (or a pass of a few taps per sample) on top of the trips of the block
	// and of its output: 8.6 us at 64 frames, 12.8 at 128 for a stereo ten-section chain; a launch of the ordinary, time-parallel kernels costs 20 ... 26 us
	// whatever the block: the wave takes blocks of up to 128 frames (profiles/r06_ladspa_rate.txt)
	// ... 256 where the block is one systolic pass and nothing per tap (19 us there against 26 - 30)
	bool taps = false;
	for (int k = 0; k < rp.n_pass; ++k) taps |= rp.pass[k].kind == RES_PASS_FIR;
	max_work = (!taps && rp.n_casc <= 1) ? 256 : 128;
	ready = true;
	return true;
}

bool Resident::open()
{
	void *c = nullptr, *mo = nullptr, *mi = nullptr;
	const size_t in_bytes = (size_t) (1 + RESIDENT_UNITS) * sizeof(ResidentUnit), out_bytes = (size_t) RESIDENT_UNITS * sizeof(ResidentUnit);
	auto fail = [&] {
		(void) hipGetLastError();
		if (mi) { if (in_device) (void) hipFree(mi); else (void) hipHostFree(mi); }
		if (mo) (void) hipHostFree(mo);
		if (c) (void) hipHostFree(c);
		ctl = nullptr; mail_in = mail_out = nullptr; st = nullptr;
		return false;
	};
	if (hipHostMalloc(&c, sizeof(ResidentCtl), hipHostMallocCoherent) != hipSuccess) { c = nullptr; return fail(); }
	if (hipHostMalloc(&mo, out_bytes, hipHostMallocCoherent) != hipSuccess) { mo = nullptr; return fail(); }
	memset(c, 0, sizeof(ResidentCtl));
	memset(mo, 0, out_bytes);
	// the request mailbox in device memory where the CPU can store into it (every byte of device memory visible through the BAR): uncached on the device side,
	// write-combined from here.  DSP_AMD_PLUGIN_MAILBOX=host keeps it in page-locked host memory (what a device without a large BAR gets anyway)
	static const bool want_device = [] { const char *v = getenv("DSP_AMD_PLUGIN_MAILBOX"); return !v || strcmp(v, "host") != 0; }();
	int dev = 0, large_bar = 0;
	(void) hipGetDevice(&dev);
	if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) != hipSuccess) { (void) hipGetLastError(); large_bar = 0; }
	in_device = false;
	if (want_device && large_bar && hipExtMallocWithFlags(&mi, in_bytes, hipDeviceMallocUncached) == hipSuccess) {
		// (trust, but verify once: what the CPU stores must be what the device holds)
		in_device = true;
		volatile unsigned long long *q = static_cast<volatile unsigned long long *>(mi);
		for (size_t k = 0; k < in_bytes / 8; ++k) q[k] = 0;
		q[2] = 0x0123456789abcdefull;
		__builtin_ia32_sfence();
		unsigned long long back = 0;
		if (hipMemcpy(&back, static_cast<char *>(mi) + 16, 8, hipMemcpyDeviceToHost) != hipSuccess || back != 0x0123456789abcdefull) {
			(void) hipGetLastError(); (void) hipFree(mi); mi = nullptr; in_device = false;
		}
		else { q[2] = 0; __builtin_ia32_sfence(); }
	}
	else (void) hipGetLastError();
	if (!mi) {
		if (hipHostMalloc(&mi, in_bytes, hipHostMallocCoherent) != hipSuccess) { mi = nullptr; return fail(); }
		memset(mi, 0, in_bytes);
	}
	if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return fail();
	ctl = static_cast<ResidentCtl *>(c);
	mail_in = static_cast<ResidentUnit *>(mi);
	mail_out = static_cast<ResidentUnit *>(mo);
	trace_mem("ctl+", ctl, sizeof(*ctl));
	rp.ctl = ctl; rp.mail_in = mail_in; rp.mail_out = mail_out;
	log_msg(LL_VERBOSE, "info: resident small-block path: request mailbox in %s memory", in_device ? "device" : "page-locked host");
	return true;
}

bool Resident::launch()
{
	__atomic_store_n(&ctl->alive, 1u, __ATOMIC_RELEASE);
	if (!launch_cascade_resident(rp, lds, st)) { (void) hipGetLastError(); __atomic_store_n(&ctl->alive, 0u, __ATOMIC_RELEASE); return false; }
	g_plugin_counters.wave_launches.fetch_add(1, std::memory_order_relaxed);
	return true;
}

static inline void res_store_unit(ResidentUnit *u, double v, unsigned long long rq)
{
	unsigned long long b;
	memcpy(&b, &v, 8);
	// (one 16-byte store: value and word travel together)
	typedef long long res_v2 __attribute__((vector_size(16), aligned(16)));
	const res_v2 q = { (long long) b, (long long) (rq ^ b) };
	*reinterpret_cast<volatile res_v2 *>(u) = q;
}

bool Resident::serve(const double *in, ssize_t frames, double *out)
{
	if (!ctl && !open()) { off = true; g_plugin_counters.wave_off.fetch_add(1, std::memory_order_relaxed); return false; }
#ifdef RES_TIMING
	const double t_enter = res_now_us();
#endif
	const unsigned served = seq;
	++seq;
	// the low word: the frames, and for every FIR pass which half of its history is current (the ordinary kernel alternates them, conv.cpp FirDirectStage)
	unsigned low = (unsigned) frames;
	for (int k = 0; k < rp.n_pass; ++k) if (fir_phase[k] && *fir_phase[k]) low |= 1u << (16 + k);
	const unsigned long long rq = ((unsigned long long) seq << 32) | (unsigned long long) low;
	const size_t n = (size_t) frames * rp.Cin, n_out = (size_t) frames * rp.Cout;
	// the block first, the control unit behind it (write-combined stores leave the core in any order: a fence in between)
	for (size_t e = 0; e < n; ++e) res_store_unit(mail_in + 1 + e, in[e], rq);
	if (in_device) __builtin_ia32_sfence();
	res_store_unit(mail_in, 0.0, rq);
	if (in_device) __builtin_ia32_sfence();
	// (the units a lane asks for together with the control unit: what this block needed -- the next one is most likely the same size)
	rp.spec_units = (int) std::min<size_t>(4, (n + 63) / 64);
	dirty = true;
	const double t0 = res_now_us();
#ifdef RES_TIMING
	t_write_us += t0 - t_enter;
#endif
	size_t got = 0;
	bool late = false;
	for (long spins = 0; got < n_out; ++spins) {
		// the reply: every unit decodes to this request once it has arrived
		while (got < n_out) {
			const volatile ResidentUnit *u = mail_out + got;
			const unsigned long long w = __atomic_load_n(&u->w, __ATOMIC_ACQUIRE);
			double v;
			{ const unsigned long long bv = *reinterpret_cast<const volatile unsigned long long *>(&u->v); memcpy(&v, &bv, 8); if ((w ^ bv) != rq) break; }
			out[got++] = v;
		}
		if (got == n_out) break;
		if (__atomic_load_n(&ctl->alive, __ATOMIC_ACQUIRE) == 0) {
			// no wave (the first block, or the last one left): start one -- it finds the request waiting.  done0 = the last request served
			rp.done0 = served;
			if (!launch()) { late = true; break; }
		}
		__builtin_ia32_pause();
		if ((spins & 255) == 255 && res_now_us() - t0 > 20000.0) { late = true; break; }      // 20 ms: something is wrong
	}
#ifdef RES_TIMING
	t_wait_us += res_now_us() - t0;
#endif
	if (!late) return true;
	// not served in time: ask the wave to leave and wait for it (bounded by its own loop).  The block goes through a launch, on the states as the wave left
	// them -- unless the wave wrote some of the reply: then it ran the block, and what is missing of the reply is on its way
	stop();
	dirty = false;
	for (int tries = 0; got > 0 && got < n_out && tries < 1000; ++tries) {
		const volatile ResidentUnit *u = mail_out + got;
		const unsigned long long w = __atomic_load_n(&u->w, __ATOMIC_ACQUIRE), bv = *reinterpret_cast<const volatile unsigned long long *>(&u->v);
		if ((w ^ bv) == rq) { double v; memcpy(&v, &bv, 8); out[got++] = v; }
	}
	if (got == n_out) return true;
	// (a partial reply that never completes cannot happen once the wave has left behind its fence; if it did, the states have advanced: say so loudly)
	if (got > 0) log_msg(LL_ERROR, "error: resident wave left a block half answered");
	g_plugin_counters.wave_timeouts.fetch_add(1, std::memory_order_relaxed);
	if (++timeouts >= 3) {
		off = true;
		g_plugin_counters.wave_off.fetch_add(1, std::memory_order_relaxed);
		log_msg(LL_VERBOSE, "info: resident small-block path switched off for this segment (three blocks were not served in time)");
	}
	return false;
}

void Resident::quiesce()
{
	if (!dirty || !ctl) return;
	// the wave says `settled == seq` the first time it finds nothing to do behind block seq (a trip of its poll loop and a wait for its stores: a few
	// microseconds); a wave that has left fenced on its way out.  Anything else: ask it to leave and wait for it
	const double t0 = res_now_us();
	for (long spins = 0;; ++spins) {
		if (__atomic_load_n(&ctl->settled, __ATOMIC_ACQUIRE) == seq || __atomic_load_n(&ctl->alive, __ATOMIC_ACQUIRE) == 0) { dirty = false; return; }
		__builtin_ia32_pause();
		if ((spins & 255) == 255 && res_now_us() - t0 > 2000.0) break;
	}
	stop();
	dirty = false;
}

void Resident::stop()
{
	if (!ctl) return;
	if (__atomic_load_n(&ctl->alive, __ATOMIC_ACQUIRE)) {
		// (a request of its own sequence number: a wave that has served block `seq` sees a new request, one that has not sees STOP in its place -- serve() only
		// stops a wave whose block it then takes through a launch)
		res_store_unit(mail_in, 0.0, ((unsigned long long) seq << 32) | (unsigned long long) RESIDENT_STOP);
		if (in_device) __builtin_ia32_sfence();
		(void) hipStreamSynchronize(st);
		(void) hipGetLastError();
	}
}

Resident::~Resident()
{
	stop();
#ifdef RES_TIMING
	if (ctl && ctl->pad[3]) {
		const double n = ctl->pad[3];
		fprintf(stderr, "resident timing: %.0f blocks; wave: mailbox -> LDS %.2f us, ops %.2f us, reply stores issued %.2f us, polls between blocks %.1f; host: block written %.2f us, reply complete %.2f us after that\n",
		        n, ctl->pad[0] / n * 0.01, ctl->pad[1] / n * 0.01, ctl->pad[2] / n * 0.01, ctl->pad[4] / n, t_write_us / n, t_wait_us / n);
	}
#endif
	if (st) { (void) hipStreamSynchronize(st); (void) hipStreamDestroy(st); }
	if (mail_in) { if (in_device) (void) hipFree(mail_in); else (void) hipHostFree(mail_in); }
	if (mail_out) (void) hipHostFree(mail_out);
	if (ctl) { trace_mem("ctl-", ctl, sizeof(*ctl)); (void) hipHostFree(ctl); }
}

static sample_t *plugin_run(struct effect *e, ssize_t *frames, sample_t *ibuf, sample_t *obuf)
{
	Node *n = node_of(e);
	if (!n || !ensure_segment(e, n)) {
		// no error channel in run() (effect.h:47): degrade to silence of the right shape and keep logging
		const ssize_t f = *frames;
		memset(obuf, 0, (size_t) f * e->ostream.channels * sizeof(sample_t));
		return obuf;
	}
	Segment &sg = *n->seg;
	if (e != sg.members.front()) return ibuf;      // the head already produced the segment's output into this buffer
	sg.touched = true;
	sample_t *dst = sg.in_place ? ibuf : obuf;
	ssize_t done = 0, produced = 0;
	const ssize_t total = *frames;
	// everything of one call is queued on the null stream in order (copy in, stages, copy out, next chunk ...) and waited
	// for once; with registered host buffers the copies are asynchronous DMA, with pageable ones they simply block
	const size_t in_bytes = (size_t) total * sg.ch_in * sizeof(double);
	const size_t out_bytes = (size_t) sg.pipe->max_out_frames(total) * sg.ch_out * sizeof(double);
	// The small-block rule (stated, not worked around: this library has no host path): a block costs a launch round trip of 23 ... 27 us whatever
	// its size (docs/history.md section 6), which the reference's own loop undercuts below about 20000 channel-sample-sections per block (64 frames
	// x 2 ch x 10 biquads: 3 us there).  A segment that keeps being driven below that says so once, at verbose level.
	++sg.calls;
	if ((double) total * sg.ch_in * std::max<size_t>(1, sg.members.size()) < 20000.0) ++sg.small_calls;
	if (!sg.advised && sg.calls == 64 && sg.small_calls > 48) {
		sg.advised = true;
		log_msg(LL_VERBOSE, "%s: info: blocks of %zd frames x %d ch through %zu effects: below about 20000 channel-sample-effects per block a device round trip "
		        "(about 25 us per block) is slower than the host's own loop; larger blocks (-b) or more channels per chain amortise it", e->name, total, sg.ch_in, sg.members.size());
	}
	if (sg.resident && sg.resident->takes(total) && sg.resident->serve(ibuf, total, dst)) {
		// (one cascade, a block the resident wave finishes sooner than a launch: no launch at all -- the block goes from the host's buffer into the
		// wave's mailbox and its output from the reply mailbox into the host's buffer)
		g_plugin_counters.wave_blocks.fetch_add(1, std::memory_order_relaxed);
		return dst;
	}
	// (the ordinary kernels are about to work on the cascade's states: a wave that has served blocks is asked to leave first -- it does not fence per block)
	if (sg.resident) sg.resident->quiesce();
	if (total <= sg.pipe_frames && sg.mapped.fits(in_bytes, out_bytes)) {
		// small block: the kernels work on the mapped staging buffers themselves
		memcpy(sg.mapped.in, ibuf, in_bytes);
		g_plugin_counters.mapped_blocks.fetch_add(1, std::memory_order_relaxed);
		const ssize_t f = sg.pipe->run(sg.mapped.in, total, sg.mapped.out, (ssize_t) (sg.mapped.bytes / (sg.ch_out * sizeof(double))), nullptr);
		(void) sg.mapped.wait_block(nullptr);
		if (f > 0) memcpy(dst, sg.mapped.out, (size_t) f * sg.ch_out * sizeof(double));
		*frames = f < 0 ? 0 : f;
		return dst;
	}
	g_plugin_counters.copied_blocks.fetch_add(1, std::memory_order_relaxed);
	// larger blocks: this library's page-locked staging buffers when the block is several pipeline calls long, else pageable copies
	{
		const ssize_t chunk = sg.pipe_frames;
		// (one chunk alone gains nothing: the thread's two copies simply add to the call -- 1.45 against 1.24 ms at 4 MB -- so those keep the
		// copy commands; from two chunks on the copies hide behind the GPU's work: 8 ch x 2^20 frames 168 -> 469 Msamples/s)
		if (total > chunk && sg.staged.ensure((size_t) std::min(total, chunk) * sg.ch_in * sizeof(double), (size_t) sg.pipe->max_out_frames(std::min(total, chunk)) * sg.ch_out * sizeof(double))) {
			// (dst may be ibuf: every chunk's input has been copied to the staging buffer before its output comes back, and a chain that
			// works in place does not make more frames than it takes)
			const ssize_t f = sg.staged.run(ibuf, total, chunk, sg.ch_in, dst, (ssize_t) 1 << 40, sg.ch_out, sg.d_in.p, sg.d_out.p, nullptr,
			                                [&](const double *di, ssize_t nb, double *dout) { return sg.pipe->run(di, nb, dout, sg.out_cap_frames, nullptr); });
			*frames = f < 0 ? 0 : f;
			return dst;
		}
	}
	while (done < total) {
		const ssize_t nb = std::min<ssize_t>(total - done, sg.pipe_frames);
		if (!hip_ok(hipMemcpyAsync(sg.d_in.p, ibuf + done * sg.ch_in, (size_t) nb * sg.ch_in * sizeof(double), hipMemcpyHostToDevice, nullptr), "H2D")) break;
		const ssize_t f = sg.pipe->run(sg.d_in.as<double>(), nb, sg.d_out.as<double>(), sg.out_cap_frames, nullptr);
		if (f < 0) break;
		if (f > 0 && !hip_ok(hipMemcpyAsync(dst + produced * sg.ch_out, sg.d_out.p, (size_t) f * sg.ch_out * sizeof(double), hipMemcpyDeviceToHost, nullptr), "D2H")) break;
		produced += f;
		done += nb;
	}
	(void) hip_ok(hipStreamSynchronize(nullptr), "sync");
	*frames = produced;
	return dst;
}

static sample_t *plugin_run_noop(struct effect *, ssize_t *, sample_t *ibuf, sample_t *)
{
	return ibuf;   // delay.c:98-101: integer delays are realised by the host's align effect
}

static sample_t *plugin_drain2(struct effect *e, ssize_t *frames, sample_t *buf1, sample_t *buf2)
{
	// the host flushes every rate changer in chain order and runs what comes out through the effects behind it
	// (effects_chain.c:1199-1217); the segment's pipeline does the same for its fused stages, the members behind
	// this effect pass the result through
	Node *n = node_of(e);
	if (!n || !n->seg || !n->seg->pipe) { *frames = -1; return buf1; }
	Segment &sg = *n->seg;
	if (sg.resident) sg.resident->quiesce();
	const ssize_t want = std::min<ssize_t>(*frames, sg.pipe_frames);
	const ssize_t f = sg.pipe->drain2(want, sg.d_out.as<double>(), sg.out_cap_frames, nullptr);
	if (f < 0) { *frames = -1; return buf1; }
	if (f > 0) {
		(void) hip_ok(hipMemcpy(buf2, sg.d_out.p, (size_t) f * sg.ch_out * sizeof(double), hipMemcpyDeviceToHost), "D2H");
	}
	*frames = f;
	return buf2;
}

static void plugin_reset(struct effect *e)
{
	Node *n = node_of(e);
	// stateless effects carry no reset callback (gain.c, remix.c, st2ms.c set none) and may head a segment: whichever
	// member with state the host resets first clears the whole segment, the others find it clean
	if (n && n->seg && n->seg->pipe && n->seg->touched) {
		if (n->seg->resident) n->seg->resident->quiesce();
		n->seg->pipe->reset(nullptr);
		(void) hipStreamSynchronize(nullptr);
		n->seg->touched = false;
	}
}

static void plugin_destroy(struct effect *e)
{
	Node *n = node_of(e);
	if (n) {
		if (n->seg && n->seg->resident) n->seg->resident->stop();      // (the device-wide wait below would otherwise sit out the wave's lifetime)
		(void) hipDeviceSynchronize();
		if (n->seg) {
			// the other members keep the segment alive but must not touch this effect any more
			for (struct effect *&m : n->seg->members) if (m == e) m = nullptr;
		}
		delete n;
	}
	e->data = nullptr;
	free(e->channel_selector);
	e->channel_selector = nullptr;
}

static int plugin_merge(struct effect *dest, struct effect *src)
{
	if (dest->merge != src->merge) return 0;
	Node *d = node_of(dest), *s = node_of(src);
	if (!d || !s || d->seg || s->seg) return 0;     // merge may only happen before the first run()
	if (!merge_specs(*d->spec, *s->spec)) return 0;
	if (d->spec->kind == Kind::Biquad)
		memcpy(dest->channel_selector, d->spec->sel.data(), d->spec->sel.size());
	return 1;
}

static void plugin_drain_samples(struct effect *e, ssize_t *samples);

static int plugin_prepare(struct effect *e)
{
	Node *n = node_of(e);
	if (!n || n->seg) return 0;
	if (n->spec->kind == Kind::Delay) {
		bool noop = true;
		if (!delay_prepare(*n->spec, &noop)) return 1;
		e->run = noop ? plugin_run_noop : plugin_run;               // delay.c:201-202
		if (!noop) { e->merge = nullptr; e->drain_samples = plugin_drain_samples; }
		if (e->channel_selector) memcpy(e->channel_selector, n->spec->sel.data(), n->spec->sel.size());
		return 0;
	}
	if (!riir_prepare(*n->spec)) return 1;
	if (e->channel_selector) memcpy(e->channel_selector, n->spec->sel.data(), n->spec->sel.size());
	return 0;
}

static void plugin_drain_samples(struct effect *e, ssize_t *samples)
{
	Node *n = node_of(e);
	if (!n) return;
	const Spec &sp = *n->spec;
	for (int k = 0; k < sp.ch_out; ++k) {
		if (sp.kind == Kind::Align) samples[k] += sp.delay[k];                                  // align.c:77-82
		else if (sp.frac_delay) samples[k] += sp.fd_ap_n[k];                                    // delay.c:105-110
		else if (!sp.ch_latency.empty()) { if (sp.sel[k]) samples[k] += sp.ch_latency[k]; }     // reverse_iir.c:234-239
		else if (sp.sel[k]) samples[k] += sp.latency + sp.T - 1;                                // fir.c:180-187, fir_p.c:235-240
	}
}

static void plugin_channel_offsets(struct effect *e, ssize_t *latency, ssize_t *req_delay)
{
	Node *n = node_of(e);
	if (!n) return;
	const Spec &sp = *n->spec;
	for (int k = 0; k < sp.ch_in; ++k) {
		if (sp.kind == Kind::Delay || sp.frac_delay) req_delay[k] += sp.delay[k];               // delay.c:142-147
		else if (!sp.ch_latency.empty()) { if (sp.sel[k]) req_delay[k] -= sp.ch_latency[k]; }   // reverse_iir.c:275-280
		else if (sp.sel[k]) { latency[k] += sp.latency; req_delay[k] -= sp.ref; }               // fir.c:208-217
	}
}

static void plugin_channel_deps(struct effect *e, char **deps)
{
	Node *n = node_of(e);
	if (!n) return;
	for (int k = 0; k < n->spec->ch_out; ++k)
		memcpy(deps[k], n->spec->remix[k].data(), n->spec->ch_in);                              // remix.c:116-121
}

#define BQ_PLOT_FMT "%.15e+%.15e*exp(-j*w)+%.15e*exp(-2.0*j*w))/(1.0+%.15e*exp(-j*w)+%.15e*exp(-2.0*j*w)"   /* biquad.h:94 */
#define BQ_PLOT_ARGS(c) (c)[0], (c)[1], (c)[2], (c)[3], (c)[4]

// one FIR channel the way fir.c:72-89 / :163-178 / fir_p.c:209-233 print it: the reference walks its zero-padded working
// length (a power of two, the FFT length, 32 + the partition groups), so the same number of terms is printed here
static void plot_fir_channel(const Spec &sp, int k, int i, int f)
{
	ssize_t terms = sp.T;
	if (sp.riir.empty()) {
		if (sp.kind == Kind::FirDirect) { terms = 1; while (terms < sp.T) terms <<= 1; }           // fir.c:253-255
		else if (sp.conv_mode == CONV_LATENCY_LEN) terms = next_fast_fftw_len(sp.T);              // fir.c:296
		else terms = fir_p_planned_len(sp.T, sp.max_part_len);
	}
	printf("H%d_%d(w)=(abs(w)<=pi)?exp(-j*w*%zd)*(0.0", k, i, -sp.ref);
	for (ssize_t n = 0; n < terms; ++n)
		printf("+exp(-j*w*%zd)*%.15e", n, n < sp.T ? sp.taps[(size_t) n * sp.fch + f] : 0.0);
	puts("):0/0");
}

static void plugin_plot(struct effect *e, int i)
{
	Node *n = node_of(e);
	if (!n) return;
	const Spec &sp = *n->spec;
	const int fs = sp.fs_out;
	switch (sp.kind) {
	case Kind::Remix:                                                     // remix.c:103-114
		for (int k = 0; k < sp.ch_out; ++k) {
			printf("H%d_%d(w)=0.0", k, i);
			for (int c = 0; c < sp.ch_in; ++c)
				if (sp.remix[k][c]) printf("+Ht%d_%d(w*%d/2.0/pi)", c, i, fs);
			putchar('\n');
		}
		return;
	case Kind::Mix: {                                                     // st2ms.c:56-70: the pair is the two selected channels
		int c0 = -1, c1 = -1;
		for (int k = 0; k < sp.ch_in; ++k) if (sp.sel[k]) { if (c0 < 0) c0 = k; else if (c1 < 0) c1 = k; }
		const double scale = (sp.name == "ms2st") ? 1.0 : 0.5;
		for (int k = 0; k < sp.ch_out; ++k) {
			if (k == c0 || k == c1)
				printf("H%d_%d(w)=(Ht%d_%d(w*%d/2.0/pi)%cHt%d_%d(w*%d/2.0/pi))*%g\n", k, i, c0, i, fs, k == c0 ? '+' : '-', c1, i, fs, scale);
			else printf("H%d_%d(w)=Ht%d_%d(w*%d/2.0/pi)\n", k, i, k, i, fs);
		}
		return;
	}
	case Kind::Crossfeed:                                                 // crossfeed.c:61-83
		for (int k = 0; k < sp.ch_out; ++k) {
			if (k == sp.xf_c0 || k == sp.xf_c1) {
				const int other = (k == sp.xf_c0) ? sp.xf_c1 : sp.xf_c0;
				printf("H%d_%d(w)=(abs(w)<=pi)?%.15e*Ht%d_%d(w*%d/2.0/pi)", k, i, sp.xf_direct, k, i, fs);
				printf("+%.15e*Ht%d_%d(w*%d/2.0/pi)*(" BQ_PLOT_FMT ")", sp.xf_cross, other, i, fs, BQ_PLOT_ARGS(sp.xf_lp));
				printf("+%.15e*Ht%d_%d(w*%d/2.0/pi)*(" BQ_PLOT_FMT ")", sp.xf_cross, k, i, fs, BQ_PLOT_ARGS(sp.xf_hp));
				puts(":0/0");
			}
			else printf("H%d_%d(w)=Ht%d_%d(w*%d/2.0/pi)\n", k, i, k, i, fs);
		}
		return;
	case Kind::Align: case Kind::Add:                                     // effect_plot_noop (effect.c:98-102; align.c:121, gain.c:122)
		for (int k = 0; k < sp.ch_in; ++k) printf("H%d_%d(f)=1.0\n", k, i);
		return;
	case Kind::FirDirect: case Kind::Conv: {
		int f = 0;
		for (int k = 0; k < sp.ch_out; ++k) {
			if (!sp.sel[k]) { printf("H%d_%d(w)=1.0\n", k, i); continue; }
			if (!sp.riir.empty()) {
				// reverse_iir.c:178-212: the pole / residue form of the designed channel (the host calls prepare() first; a
				// channel still waiting for its design is printed from nothing better than the identity)
				if (k < (int) sp.riir_plot.size() && !sp.riir_plot[k].empty())
					printf("H%d_%d(w)=(abs(w)<=pi)?1.0%s*exp(%zd*j*w):0/0\n", k, i, sp.riir_plot[k].c_str(), sp.ch_latency[k]);
				else printf("H%d_%d(w)=1.0\n", k, i);
			}
			else plot_fir_channel(sp, k, i, sp.fch > 1 ? f : 0);
			++f;
		}
		return;
	}
	default: break;
	}
	for (int k = 0; k < sp.ch_out; ++k) {
		if (sp.kind == Kind::Biquad && sp.frac_delay) {                   // delay.c:84-104 (orders above 2: the same all-pass as a product of sections)
			printf("H%d_%d(w)=exp(-j*w*%zd)", k, i, sp.delay[k]);
			if (sp.sel[k]) {
				printf("*((abs(w)<=pi)?(" BQ_PLOT_FMT ")", BQ_PLOT_ARGS(sp.bq[k]));
				for (size_t m = 0; m < sp.bq_more.size(); ++m)
					if (sp.sel_more[m][k]) printf("*(" BQ_PLOT_FMT ")", BQ_PLOT_ARGS(sp.bq_more[m][k]));
				printf(":0/0)");
			}
			putchar('\n');
		}
		else if (sp.kind == Kind::Biquad && sp.sel[k])                    // biquad.c:325-336
			printf("H%d_%d(w)=(abs(w)<=pi)?(" BQ_PLOT_FMT "):0/0\n", k, i, BQ_PLOT_ARGS(sp.bq[k]));
		else if (sp.kind == Kind::Gain) printf("H%d_%d(w)=%.15e\n", k, i, sp.vec[k]);           // gain.c:45-50
		else if (sp.kind == Kind::Delay) printf("H%d_%d(w)=exp(-j*w*%zd)\n", k, i, sp.delay[k]);
		else printf("H%d_%d(w)=1.0\n", k, i);
	}
}

struct effect *make_effect(SpecPtr spec, bool noop)
{
	struct effect *e = static_cast<struct effect *>(calloc(1, sizeof(struct effect)));
	if (!e) { set_error("out of memory"); return nullptr; }
	Node *n = new Node;
	const Spec &sp = *spec;
	e->istream.fs = sp.fs_in; e->istream.channels = sp.ch_in;
	e->ostream.fs = sp.fs_out; e->ostream.channels = sp.ch_out;
	e->channel_selector = static_cast<char *>(calloc(sp.ch_in ? sp.ch_in : 1, 1));
	if (e->channel_selector) memcpy(e->channel_selector, sp.sel.data(), sp.sel.size());
	e->flags = sp.flags;
	e->destroy = plugin_destroy;
	e->data = n;
	if (!noop) {
		e->run = (sp.kind == Kind::Delay) ? plugin_run_noop : plugin_run;
		// reset only where the reference's effect has one (gain.c / remix.c / st2ms.c: none)
		if (sp.kind != Kind::Gain && sp.kind != Kind::Add && sp.kind != Kind::Remix && sp.kind != Kind::Mix) e->reset = plugin_reset;
		switch (sp.kind) {
		case Kind::Gain: case Kind::Add: e->merge = plugin_merge; e->plot = plugin_plot; break;
		case Kind::Biquad: e->merge = plugin_merge; e->plot = plugin_plot; break;
		case Kind::Delay: e->merge = plugin_merge; e->plot = plugin_plot; e->channel_offsets = plugin_channel_offsets; e->prepare = plugin_prepare; break;
		case Kind::Align: e->drain_samples = plugin_drain_samples; e->plot = plugin_plot; break;
		case Kind::Remix: case Kind::Mix: case Kind::Crossfeed: e->channel_deps = plugin_channel_deps; e->plot = plugin_plot; break;
		case Kind::FirDirect: case Kind::Conv:
			e->drain_samples = plugin_drain_samples;
			e->channel_offsets = plugin_channel_offsets;
			if (sp.conv_mode != CONV_ZITA_EQUIV) e->plot = plugin_plot;       // zita_convolver.cpp sets none
			if (sp.riir_pending) { e->merge = plugin_merge; e->prepare = plugin_prepare; }
			break;
		case Kind::Resample: e->drain2 = plugin_drain2; break;
		}
	}
	n->spec = std::move(spec);
	e->name = n->spec->name.c_str();
	return e;
}

struct effect *make_align_effect(int fs, int channels, const std::vector<ssize_t> &len, ssize_t discard)
{
	return make_effect(make_align_spec(fs, channels, len, discard), false);
}

}  // namespace dspamd

// ------------------------------------------------------------------ exported entry points

using namespace dspamd;

extern "C" {

struct effect *biquad_effect_init(const struct effect_info *ei, const struct stream_info *is, const char *sel, const char *dir, int argc, const char *const *argv)
{
	(void) dir;
	bool rev = false;
	SpecPtr s = parse_biquad(ei->effect_number, is, sel, argc, argv, &rev);
	return s ? make_effect(std::move(s), false) : nullptr;
}

struct effect *gain_effect_init(const struct effect_info *ei, const struct stream_info *is, const char *sel, const char *dir, int argc, const char *const *argv)
{
	(void) dir;
	SpecPtr s = parse_gain(ei->effect_number, is, sel, argc, argv);
	return s ? make_effect(std::move(s), false) : nullptr;
}

struct effect *remix_effect_init(const struct effect_info *ei, const struct stream_info *is, const char *sel, const char *dir, int argc, const char *const *argv)
{
	(void) ei; (void) dir;
	SpecPtr s = parse_remix(is, sel, argc, argv);
	return s ? make_effect(std::move(s), false) : nullptr;
}

struct effect *st2ms_effect_init(const struct effect_info *ei, const struct stream_info *is, const char *sel, const char *dir, int argc, const char *const *argv)
{
	(void) dir;
	SpecPtr s = parse_st2ms(ei->effect_number, is, sel, argc, argv);
	return s ? make_effect(std::move(s), false) : nullptr;
}

struct effect *crossfeed_effect_init(const struct effect_info *ei, const struct stream_info *is, const char *sel, const char *dir, int argc, const char *const *argv)
{
	(void) ei; (void) dir;
	SpecPtr s = parse_crossfeed(is, sel, argc, argv);
	return s ? make_effect(std::move(s), false) : nullptr;
}

struct effect *delay_effect_init(const struct effect_info *ei, const struct stream_info *is, const char *sel, const char *dir, int argc, const char *const *argv)
{
	(void) ei; (void) dir;
	bool noop = false;
	SpecPtr s = parse_delay(is, sel, argc, argv, &noop);
	return s ? make_effect(std::move(s), noop) : nullptr;
}

struct effect *delay_effect_init_int(const char *name, const struct stream_info *is, const char *sel, ssize_t samples_int)
{
	bool noop = false;
	SpecPtr s = make_delay_spec(name, is, sel, samples_int, &noop);
	return s ? make_effect(std::move(s), noop) : nullptr;
}

struct effect *delay_effect_init_frac(const char *name, const struct stream_info *is, const char *sel, double samples_frac, int fd_ap_n)
{
	bool noop = false;
	SpecPtr s = make_frac_delay_spec(name, is, sel, samples_frac, fd_ap_n, &noop);
	return s ? make_effect(std::move(s), noop) : nullptr;
}

struct effect *fir_effect_init(const struct effect_info *ei, const struct stream_info *is, const char *sel, const char *dir, int argc, const char *const *argv)
{
	SpecPtr s = parse_fir(ei->name, false, is, sel, dir, argc, argv);
	return s ? make_effect(std::move(s), false) : nullptr;
}

struct effect *fir_p_effect_init(const struct effect_info *ei, const struct stream_info *is, const char *sel, const char *dir, int argc, const char *const *argv)
{
	SpecPtr s = parse_fir(ei->name, true, is, sel, dir, argc, argv);
	return s ? make_effect(std::move(s), false) : nullptr;
}

struct effect *zita_convolver_effect_init(const struct effect_info *ei, const struct stream_info *is, const char *sel, const char *dir, int argc, const char *const *argv)
{
	(void) ei;
	SpecPtr s = parse_zita(is, sel, dir, argc, argv);
	return s ? make_effect(std::move(s), false) : nullptr;
}

struct effect *hilbert_effect_init(const struct effect_info *ei, const struct stream_info *is, const char *sel, const char *dir, int argc, const char *const *argv)
{
	(void) ei; (void) dir;
	SpecPtr s = parse_hilbert(is, sel, argc, argv);
	return s ? make_effect(std::move(s), false) : nullptr;
}

struct effect *resample_effect_init(const struct effect_info *ei, const struct stream_info *is, const char *sel, const char *dir, int argc, const char *const *argv)
{
	(void) ei; (void) dir;
	bool noop = false;
	SpecPtr s = parse_resample(is, sel, argc, argv, &noop);
	return s ? make_effect(std::move(s), noop) : nullptr;
}

struct effect *fir_effect_init_with_filter(const struct effect_info *ei, const struct stream_info *is, const char *sel, sample_t *filter_data,
	int filter_channels, ssize_t filter_frames, ssize_t ref, int force_direct)
{
	SpecPtr s = make_fir_spec(ei->name, is, sel, filter_data, filter_channels, filter_frames, ref, CONV_LATENCY_LEN, force_direct, 0);
	return s ? make_effect(std::move(s), false) : nullptr;
}

struct effect *fir_p_effect_init_with_filter(const struct effect_info *ei, const struct stream_info *is, const char *sel, sample_t *filter_data,
	int filter_channels, ssize_t filter_frames, ssize_t ref, int max_part_len)
{
	SpecPtr s = make_fir_spec(ei->name, is, sel, filter_data, filter_channels, filter_frames, ref, CONV_ZERO_LATENCY, 0, 0);
	if (s) s->max_part_len = max_part_len;
	return s ? make_effect(std::move(s), false) : nullptr;
}

struct effect *zita_convolver_effect_init_with_filter(const struct effect_info *ei, const struct stream_info *is, const char *sel, sample_t *filter_data,
	int filter_channels, ssize_t filter_frames, ssize_t ref, int min_part_len, int max_part_len)
{
	(void) max_part_len;
	SpecPtr s = make_fir_spec(ei->name, is, sel, filter_data, filter_channels, filter_frames, ref, CONV_ZITA_EQUIV, 0, min_part_len);
	return s ? make_effect(std::move(s), false) : nullptr;
}

}  // extern "C"

// ------------------------------------------------------------------ registry (mirror of effect.c:46-76 for these effects)

namespace dspamd {

#define FIR_OPTS "[-a[offset[s|m|S]]] [input_options]"
#define FIR_FILTER "[file:][~/]filter_path|coefs:list[/list...]"

static const effect_info g_effects[] = {
	{ "lowpass_1",          "[-r[thresh]] f0[k]",                             biquad_effect_init, DSPAMD_BIQUAD_LOWPASS_1 },
	{ "highpass_1",         "[-r[thresh]] f0[k]",                             biquad_effect_init, DSPAMD_BIQUAD_HIGHPASS_1 },
	{ "allpass_1",          "[-r[thresh]] f0[k]",                             biquad_effect_init, DSPAMD_BIQUAD_ALLPASS_1 },
	{ "lowshelf_1",         "[-r[thresh]] f0[k] gain",                        biquad_effect_init, DSPAMD_BIQUAD_LOWSHELF_1 },
	{ "highshelf_1",        "[-r[thresh]] f0[k] gain",                        biquad_effect_init, DSPAMD_BIQUAD_HIGHSHELF_1 },
	{ "lowpass_1p",         "[-r[thresh]] f0[k]",                             biquad_effect_init, DSPAMD_BIQUAD_LOWPASS_1P },
	{ "lowpass",            "[-r[thresh]] f0[k] width[q|o|h|k]",              biquad_effect_init, DSPAMD_BIQUAD_LOWPASS },
	{ "highpass",           "[-r[thresh]] f0[k] width[q|o|h|k]",              biquad_effect_init, DSPAMD_BIQUAD_HIGHPASS },
	{ "bandpass_skirt",     "[-r[thresh]] f0[k] width[q|o|h|k]",              biquad_effect_init, DSPAMD_BIQUAD_BANDPASS_SKIRT },
	{ "bandpass_peak",      "[-r[thresh]] f0[k] width[q|o|h|k]",              biquad_effect_init, DSPAMD_BIQUAD_BANDPASS_PEAK },
	{ "notch",              "[-r[thresh]] f0[k] width[q|o|h|k]",              biquad_effect_init, DSPAMD_BIQUAD_NOTCH },
	{ "allpass",            "[-r[thresh]] f0[k] width[q|o|h|k]",              biquad_effect_init, DSPAMD_BIQUAD_ALLPASS },
	{ "eq",                 "[-r[thresh]] f0[k] width[q|o|h|k] gain",         biquad_effect_init, DSPAMD_BIQUAD_PEAK },
	{ "lowshelf",           "[-r[thresh]] f0[k] width[q|s|d|o|h|k] gain",     biquad_effect_init, DSPAMD_BIQUAD_LOWSHELF },
	{ "highshelf",          "[-r[thresh]] f0[k] width[q|s|d|o|h|k] gain",     biquad_effect_init, DSPAMD_BIQUAD_HIGHSHELF },
	{ "lowpass_transform",  "[-r[thresh]] fz[k] width_z[q] fp[k] width_p[q]", biquad_effect_init, DSPAMD_BIQUAD_LOWPASS_TRANSFORM },
	{ "highpass_transform", "[-r[thresh]] fz[k] width_z[q] fp[k] width_p[q]", biquad_effect_init, DSPAMD_BIQUAD_HIGHPASS_TRANSFORM },
	{ "linkwitz_transform", "[-r[thresh]] fz[k] width_z[q] fp[k] width_p[q]", biquad_effect_init, DSPAMD_BIQUAD_HIGHPASS_TRANSFORM },
	{ "deemph",             "[-r[thresh]]",                                   biquad_effect_init, DSPAMD_BIQUAD_DEEMPH },
	{ "biquad",             "[-r[thresh]] b0 b1 b2 a0 a1 a2",                 biquad_effect_init, DSPAMD_BIQUAD_BIQUAD },
	{ "gain",               "gain_dB",                                        gain_effect_init, DSPAMD_GAIN_GAIN },
	{ "mult",               "multiplier",                                     gain_effect_init, DSPAMD_GAIN_MULT },
	{ "add",                "value",                                          gain_effect_init, DSPAMD_GAIN_ADD },
	{ "remix",              "channel_selector|. ...",                         remix_effect_init, 0 },
	{ "st2ms",              "",                                               st2ms_effect_init, DSPAMD_ST2MS_ST2MS },
	{ "ms2st",              "",                                               st2ms_effect_init, DSPAMD_ST2MS_MS2ST },
	{ "crossfeed",          "f0[k] separation",                               crossfeed_effect_init, 0 },
	{ "delay",              "[-f[order]] [-m|M depth[s|m|S|%]] [-b bw[k]] [-q quality] delay[s|m|S]", delay_effect_init, 0 },
	{ "resample",           "[bandwidth] fs[k]|x{mult}|/{div}",                        resample_effect_init, 0 },
	{ "fir",                FIR_OPTS " " FIR_FILTER,                          fir_effect_init, 0 },
	{ "fir_p",              FIR_OPTS " [max_part_len] " FIR_FILTER,           fir_p_effect_init, 0 },
	{ "zita_convolver",     FIR_OPTS " [min_part_len [max_part_len]] " FIR_FILTER, zita_convolver_effect_init, 0 },
	{ "hilbert",            "[-pzc] [-a angle] taps",                         hilbert_effect_init, 0 },
};

const effect_info *registry_table(int *n)
{
	*n = (int) (sizeof(g_effects) / sizeof(g_effects[0]));
	return g_effects;
}

const effect_info *registry_lookup(const char *name)
{
	for (const effect_info &ei : g_effects)
		if (strcmp(ei.name, name) == 0) return &ei;
	return nullptr;
}

}  // namespace dspamd
