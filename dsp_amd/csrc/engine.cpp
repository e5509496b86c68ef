// engine.cpp -- pipeline compiler and the simple stages (cascade, remix, align/delay).
// The FFT convolution and resample stages live in conv.cpp.
#include "engine.h"
#include <algorithm>
#include <map>
#include <mutex>
#include "stages.h"
#include "fft_params.h"
#include "pcm_params.h"
#include <cmath>
#include <ctime>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <atomic>
#include <condition_variable>
#include <thread>
#include <vector>
#include <sched.h>
#include <unistd.h>

namespace dspamd {

bool hip_ok(hipError_t e, const char *what)
{
	if (e == hipSuccess) return true;
	set_error("HIP error in %s: %s", what, hipGetErrorString(e));
	return false;
}

void grant_dynamic_lds(const void *kernel, size_t bytes)
{
	static std::mutex mu;
	static std::map<std::pair<const void *, int>, size_t> granted;
	int dev = 0;
	(void) hipGetDevice(&dev);
	std::lock_guard<std::mutex> lock(mu);
	size_t &g = granted[std::make_pair(kernel, dev)];
	if (bytes <= g) return;
	if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes) == hipSuccess) g = bytes;
	else (void) hipGetLastError();   // the launch itself will report the failure
}

// A build with -DDSP_AMD_TRACE_MEM (make TRACE_MEM=1): DSP_AMD_TRACE_MEM=<file> gets one line per device / page-locked allocation and release of this
// library (what, address, bytes, microseconds) -- the map a GPU memory fault's address is looked up in.  Not in the product build.
void trace_mem(const char *what, const void *p, size_t n)
{
#ifndef DSP_AMD_TRACE_MEM
	(void) what; (void) p; (void) n;
#else
	static FILE *f = [] { const char *e = getenv("DSP_AMD_TRACE_MEM"); FILE *h = (e && *e) ? fopen(e, "a") : nullptr; if (h) setvbuf(h, nullptr, _IOLBF, 0); return h; }();
	if (!f) return;
	timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
	fprintf(f, "%s %p %zu %lld pid %d\n", what, p, n, (long long) t.tv_sec * 1000000 + t.tv_nsec / 1000, (int) getpid());
#endif
}

int device_count()
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

Profiler g_prof;

void Profiler::begin(const char *name, hipStream_t st)
{
	Rec r{ name, nullptr, nullptr };
	(void) hipEventCreate(&r.a);
	(void) hipEventCreate(&r.b);
	(void) hipEventRecord(r.a, st);
	recs.push_back(r);
}

void Profiler::end(hipStream_t st)
{
	if (!recs.empty()) (void) hipEventRecord(recs.back().b, st);
}

void Profiler::collect(std::vector<std::string> &names, std::vector<double> &ms, std::vector<long> &counts)
{
	for (Rec &r : recs) {
		float t = 0.f;
		if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
			size_t i = 0;
			for (; i < names.size(); ++i) if (names[i] == r.name) break;
			if (i == names.size()) { names.push_back(r.name); ms.push_back(0.0); counts.push_back(0); }
			ms[i] += t;
			counts[i] += 1;
		}
		(void) hipEventDestroy(r.a);
		(void) hipEventDestroy(r.b);
	}
	recs.clear();
}

Profiler::~Profiler()
{
	for (Rec &r : recs) { (void) hipEventDestroy(r.a); (void) hipEventDestroy(r.b); }
}

static const size_t CANARY = 4096, GUARD_SLACK = (size_t) 64 << 10;      // (64 KB of poison behind a buffer in mode 5: one 4096-point row of a transform)
// DSP_AMD_GUARD (debugging, read once): 3 = a page of a pattern on both sides of every device buffer, looked at when the buffer goes (writes outside a
// buffer, found where they land); 5 = every buffer that is not asked to be zero, and the slack behind every buffer, filled with 0xFF bytes (a NaN in every
// double and float; the slack: 64 KB that only this mode allocates): a kernel that USES memory nobody has written -- or what lies behind a buffer -- shows
// up as a NaN in a parity test, whatever the allocator happened to leave there.  (Modes 1 / 2 / 4 of round 5, buffers in hipMemCreate mappings of their own, are gone: memory mapped that way gave
// wrong values in large-buffer tests whatever the placement, so they said nothing about over-reads.)
static int guard_mode() { static const int m = [] { const char *e = getenv("DSP_AMD_GUARD"); return e ? atoi(e) : 0; }(); return m; }

bool DevBuf::alloc(size_t n, bool zero)
{
	release();
	if (n == 0) n = 16;
	if (guard_mode() == 3) {
		char *base = nullptr;
		if (hipMalloc((void **) &base, n + 2 * CANARY) == hipSuccess && hipMemset(base, 0xA5, n + 2 * CANARY) == hipSuccess) {
			p = base + CANARY; canary = true; site = __builtin_return_address(0);
		}
		else (void) hipGetLastError();
	}
	// (round 5 put 64 KB of slack behind every buffer as a defence against an over-read nobody had found; the fault it was meant for was not an over-read
	// -- DESIGN.md section 5 -- and the slack is gone: a buffer is what was asked for, and the descriptors the kernels address it through say so)
	const size_t slack = (guard_mode() == 5) ? GUARD_SLACK : 0;
	if (!p && !hip_ok(hipMalloc(&p, n + slack), "hipMalloc")) { p = nullptr; return false; }
	bytes = n;
	trace_mem("dev+", p, n);
	if (guard_mode() == 5 && !canary && !hip_ok(hipMemset(p, 0xFF, n + slack), "hipMemset")) return false;
	if (zero && !hip_ok(hipMemset(p, 0, n), "hipMemset")) return false;
	return true;
}

void MappedPair::alloc()
{
	static const long kb = [] { const char *v = getenv("DSP_AMD_PLUGIN_MAPPED_KB"); return v ? atol(v) : 32L; }();
	if (kb <= 0 || bytes) return;
	void *a = nullptr, *b = nullptr;
	// (coherent = fine-grained: a wave that stays on the device across blocks -- kernels_resident.hip -- must see what the host wrote a moment ago, not an L2 line)
	if (hipHostMalloc(&a, (size_t) kb << 10, hipHostMallocCoherent) == hipSuccess && hipHostMalloc(&b, (size_t) kb << 10, hipHostMallocCoherent) == hipSuccess) {
		in = static_cast<double *>(a); out = static_cast<double *>(b); bytes = (size_t) kb << 10;
		trace_mem("map+", a, bytes); trace_mem("map+", b, bytes);
		void *f = nullptr;
		if (hipHostMalloc(&f, 64, hipHostMallocDefault) == hipSuccess) { flag = static_cast<volatile unsigned *>(f); *flag = 0; }
		else (void) hipGetLastError();
		return;
	}
	(void) hipGetLastError();
	if (a) (void) hipHostFree(a);
}

MappedPair::~MappedPair()
{
	if (in) { trace_mem("map-", in, bytes); (void) hipHostFree(in); }
	if (out) { trace_mem("map-", out, bytes); (void) hipHostFree(out); }
	if (flag) (void) hipHostFree(const_cast<unsigned *>(flag));
}

namespace {
// Large host-to-pinned copies spread over the caller and three helper threads.  The block is cut into slices that the participants
// CLAIM one at a time (an atomic counter): a helper that does not get a core -- a cgroup quota, a SCHED_FIFO caller on its CPU --
// simply claims nothing and the caller copies the whole block itself; the caller only ever waits for slices that are being copied,
// first spinning, then sleeping so that a preempted helper can run.  The helpers are detached and live as long as the process:
// the library is linked -z nodelete (Makefile), so a dlclose never unmaps the code under them.  DSP_AMD_COPY_CREW=0: caller alone.
struct CopyCrew {
	static constexpr int HELPERS = 3;
	static constexpr unsigned NS = 16;       // slices per block, whatever its size (a constant: a late helper's claim is judged without reading anything of the block)
	std::mutex use;                          // one block at a time
	std::mutex m;
	std::condition_variable cv;
	std::atomic<unsigned> gen { 0 };         // one step per block handed out (what the helpers sleep on)
	std::atomic<uint64_t> next { NS };       // (block number << 32) | the next unclaimed slice of that block
	std::atomic<unsigned> done { 0 };        // slices of the current block copied
	char *dst = nullptr;
	const char *src = nullptr;
	size_t slice = 0, bytes = 0;
	int state = 0;                           // 0 not started, 1 running, -1 not available
	pid_t pid = 0;                           // the process the helpers live in (a forked child has none)
	static double now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
	// claim and copy slices until none is left.  A claim with index < NS belongs to the block whose number came with it, and that
	// block's fields stay as they are until the slice is counted done (the caller does not move on before); a helper that wakes
	// up late either finds the counter past NS or claims a slice of a later block -- which is just as valid.
	void work()
	{
		for (;;) {
			const uint64_t v = next.fetch_add(1, std::memory_order_acq_rel);
			const unsigned i = (unsigned) (v & 0xffffffffu);
			if (i >= NS) return;
			const size_t off = (size_t) i * slice;
			if (off < bytes) memcpy(dst + off, src + off, std::min(slice, bytes - off));
			done.fetch_add(1, std::memory_order_release);
		}
	}
	void helper()
	{
		unsigned seen = 0;
		for (;;) {
			// the next block of a burst follows within a few hundred microseconds: look for it that long, then sleep
			const double t0 = now_us();
			while (gen.load(std::memory_order_acquire) == seen) {
				if (now_us() - t0 > 500.0) {
					std::unique_lock<std::mutex> lk(m);
					cv.wait(lk, [&] { return gen.load(std::memory_order_acquire) != seen; });
					break;
				}
				__builtin_ia32_pause();
			}
			seen = gen.load(std::memory_order_acquire);
			work();
		}
	}
	bool start()
	{
		cpu_set_t set;
		if (sched_getaffinity(0, sizeof set, &set) != 0 || CPU_COUNT(&set) < HELPERS + 1) { state = -1; return false; }
		try {
			for (int i = 0; i < HELPERS; ++i) std::thread([this] { helper(); }).detach();
		} catch (...) { state = -1; return false; }      // (helpers already started stay asleep: gen never moves)
		state = 1;
		pid = getpid();
		return true;
	}
	void copy(void *d, const void *s, size_t n)
	{
		if (n < ((size_t) 1 << 20) || !use.try_lock()) { memcpy(d, s, n); return; }
		if (state == 0) start();
		if (state == 1 && pid != getpid()) state = -1;
		if (state != 1) { use.unlock(); memcpy(d, s, n); return; }
		// (no claim of the previous block is open: all its NS slices were counted done before copy() returned)
		dst = static_cast<char *>(d); src = static_cast<const char *>(s); bytes = n;
		slice = ((n + NS - 1) / NS + 4095) & ~(size_t) 4095;
		done.store(0, std::memory_order_relaxed);
		unsigned g;
		{
			std::lock_guard<std::mutex> lk(m);
			g = gen.load(std::memory_order_relaxed) + 1;
			next.store((uint64_t) g << 32, std::memory_order_release);      // publishes the fields above with the first claimable index
			gen.store(g, std::memory_order_release);
		}
		cv.notify_all();
		work();
		// every slice is claimed by now; the ones still in a helper's hands take microseconds -- unless that helper lost its core
		const double t0 = now_us();
		while (done.load(std::memory_order_acquire) != NS) {
			if (now_us() - t0 < 200.0) __builtin_ia32_pause();
			else { timespec ts = { 0, 50000 }; nanosleep(&ts, nullptr); }       // (sleeping, not yielding: a real-time caller must leave the CPU for the helper to finish)
		}
		use.unlock();
	}
};
CopyCrew *crew() { static CopyCrew *c = new CopyCrew; return c; }     // (never destroyed: its threads may outlive static destruction)
}  // namespace

void crew_memcpy(void *dst, const void *src, size_t bytes) { crew()->copy(dst, src, bytes); }

bool PinnedStage::ensure(size_t in_bytes, size_t out_bytes)
{
	static const bool enabled = [] { const char *e = getenv("DSP_AMD_PLUGIN_STAGE"); return !e || atoi(e) != 0; }();
	if (!enabled || off) return false;
	if (in_bytes <= in_cap && out_bytes <= out_cap) return true;
	if (std::max(in_bytes, out_bytes) > ((size_t) 64 << 20)) return false;
	if (done[0]) (void) hipDeviceSynchronize();           // (buffers that may still be in flight are about to be replaced)
	for (int i = 0; i < 2; ++i) {
		if (in[i]) { trace_mem("stage-", in[i], in_cap); (void) hipHostFree(in[i]); }
		if (out[i]) { trace_mem("stage-", out[i], out_cap); (void) hipHostFree(out[i]); }
		in[i] = out[i] = nullptr;
	}
	in_cap = out_cap = 0;
	bool ok = true;
	for (int i = 0; i < 2 && ok; ++i) {
		void *a = nullptr, *b = nullptr;
		ok = hipHostMalloc(&a, std::max<size_t>(in_bytes, 4096), hipHostMallocDefault) == hipSuccess && hipHostMalloc(&b, std::max<size_t>(out_bytes, 4096), hipHostMallocDefault) == hipSuccess;
		in[i] = static_cast<char *>(a); out[i] = static_cast<char *>(b);
		if (ok && !done[i]) ok = hipEventCreateWithFlags(&done[i], hipEventDisableTiming) == hipSuccess;
		if (ok && !copied[i]) ok = hipEventCreateWithFlags(&copied[i], hipEventDisableTiming) == hipSuccess;
	}
	if (!ok) {
		(void) hipGetLastError();
		for (int i = 0; i < 2; ++i) { if (in[i]) (void) hipHostFree(in[i]); if (out[i]) (void) hipHostFree(out[i]); in[i] = out[i] = nullptr; }
		off = true;
		return false;
	}
	in_cap = std::max<size_t>(in_bytes, 4096); out_cap = std::max<size_t>(out_bytes, 4096);
	for (int i = 0; i < 2; ++i) { trace_mem("stage+", in[i], in_cap); trace_mem("stage+", out[i], out_cap); }
	return true;
}

PinnedStage::~PinnedStage()
{
	for (int i = 0; i < 2; ++i) {
		if (in[i]) { trace_mem("stage-", in[i], in_cap); (void) hipHostFree(in[i]); }
		if (out[i]) { trace_mem("stage-", out[i], out_cap); (void) hipHostFree(out[i]); }
		if (done[i]) (void) hipEventDestroy(done[i]);
		if (copied[i]) (void) hipEventDestroy(copied[i]);
	}
}

bool MappedPair::wait_block(hipStream_t st)
{
	if (flag && !flag_off) {
		const unsigned want = ++seq;
		if (hipStreamWriteValue32(st, const_cast<unsigned *>(flag), want, 0) == hipSuccess) {
			// a small block is through in tens of microseconds: poll for at most 2 ms by the clock (a pause is 40 ... 140 cycles: counting
			// them bounds nothing), then let the runtime wait
			timespec t0;
			clock_gettime(CLOCK_MONOTONIC, &t0);
			for (long spins = 0;; ++spins) {
				if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == want) return true;       // (acquire: the block's output is read after this)
				__builtin_ia32_pause();
				if ((spins & 255) == 255) {
					timespec t1;
					clock_gettime(CLOCK_MONOTONIC, &t1);
					if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 2000000L) break;
				}
			}
			return hip_ok(hipStreamSynchronize(st), "sync");
		}
		(void) hipGetLastError();
		flag_off = true;                         // (no stream memory operations on this runtime / device: the plain wait from now on)
	}
	return hip_ok(hipStreamSynchronize(st), "sync");
}

bool DevBuf::upload(const void *src, size_t n)
{
	if (!alloc(n, false)) return false;
	return hip_ok(hipMemcpy(p, src, n, hipMemcpyHostToDevice), "hipMemcpy H2D");
}

void DevBuf::release()
{
	if (p && canary) {
		std::vector<unsigned char> h(bytes + 2 * CANARY);
		char *base = static_cast<char *>(p) - CANARY;
		if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(h.data(), base, h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
			long lo = -1, hi = -1, n_bad = 0;
			for (size_t i = 0; i < h.size(); ++i) {
				if (i >= CANARY && i < CANARY + bytes) continue;
				if (h[i] != 0xA5) { if (lo < 0) lo = (long) i; hi = (long) i; ++n_bad; }
			}
			if (n_bad) fprintf(stderr, "dsp_amd: CANARY: buffer of %zu bytes allocated at %p: %ld bytes written outside, offsets %ld ... %ld relative to its start\n",
			                   bytes, site, n_bad, lo - (long) CANARY, hi - (long) CANARY);
		}
		else (void) hipGetLastError();
		(void) hipFree(base);
		canary = false;
	}
	else if (p) { trace_mem("dev-", p, bytes); (void) hipFree(p); }
	p = nullptr;
	bytes = 0;
}

// ------------------------------------------------------------ CascadeStage

static void mat_mul(const long double a[4], const long double b[4], long double r[4])
{
	r[0] = a[0]*b[0] + a[1]*b[2];
	r[1] = a[0]*b[1] + a[1]*b[3];
	r[2] = a[2]*b[0] + a[3]*b[2];
	r[3] = a[2]*b[1] + a[3]*b[3];
}

static void fill_biquad_op(OpDesc &od, const std::array<double, 5> &c)
{
	od.kind = OP_BIQUAD;
	for (int i = 0; i < 5; ++i) od.c[i] = c[i];
	// A = [[-c3, 1], [-c4, 0]]; powers by repeated squaring in extended precision
	long double A[4] = { -(long double) c[3], 1.0L, -(long double) c[4], 0.0L };
	long double Pk[4] = { A[0], A[1], A[2], A[3] };
	for (int k = 0; k < CASCADE_NPOW; ++k) {
		for (int q = 0; q < 4; ++q) od.P[k][q] = (double) Pk[q];
		long double sq[4];
		mat_mul(Pk, Pk, sq);
		memcpy(Pk, sq, sizeof(sq));
	}
	// h[i] = first row of A^i
	long double row[2] = { 1.0L, 0.0L };
	for (int i = 0; i < CASCADE_L; ++i) {
		od.h[i][0] = (double) row[0];
		od.h[i][1] = (double) row[1];
		const long double n0 = row[0]*A[0] + row[1]*A[2], n1 = row[0]*A[1] + row[1]*A[3];
		row[0] = n0;
		row[1] = n1;
	}
}

void CascadeStage::add(const Spec &sp)
{
	std::vector<OpDesc> col(ch_in);
	for (int k = 0; k < ch_in; ++k) {
		OpDesc &od = col[k];
		memset(&od, 0, sizeof(od));
		if (sp.kind == Kind::Gain) { od.kind = OP_MUL; od.g = sp.vec[k]; }
		else if (sp.kind == Kind::Add) { od.kind = OP_ADD; od.g = sp.vec[k]; }
		else if (sp.sel[k]) fill_biquad_op(od, sp.bq[k]);
		else od.kind = OP_SKIP;
	}
	cols.push_back(std::move(col));
	names.push_back(sp.name);
}

bool CascadeStage::finalize()
{
	n_ops = (int) cols.size();
	std::vector<OpDesc> host((size_t) ch_in * n_ops);
	for (int c = 0; c < ch_in; ++c)
		for (int j = 0; j < n_ops; ++j)
			host[(size_t) c * n_ops + j] = cols[j][c];
	if (!ops.upload(host.data(), host.size() * sizeof(OpDesc))) return false;
	host_ops = host;
	// (a chain of gains alone stays bit-exact on the ordinary path; the correction pass would turn -0.0 into +0.0)
	bool has_add = false, has_section = false;
	for (const OpDesc &od : host) { if (od.kind == OP_ADD) has_add = true; if (od.kind == OP_BIQUAD) has_section = true; }
	chunk_linear = has_section && !has_add;
	// tables of the fast kernel: wave-uniform constants (scalar loads) and per-lane carry matrices
	{
		std::vector<double> tab((size_t) ch_in * n_ops * FOP_DOUBLES, 0.0), q((size_t) ch_in * n_ops * FQ_DOUBLES, 0.0);
		int lg = 0;
		while ((1 << lg) < CASCADE_L) ++lg;
		for (size_t e = 0; e < host.size(); ++e) {
			const OpDesc &od = host[e];
			double *d = &tab[e * FOP_DOUBLES];
			long long kind = od.kind;
			memcpy(&d[0], &kind, sizeof(kind));
			d[1] = od.g;
			for (int i = 0; i < 5; ++i) d[2 + i] = od.c[i];
			if (od.kind != OP_BIQUAD) continue;
			for (int k = 0; k < 5; ++k) for (int i = 0; i < 4; ++i) d[FOP_PW + 4 * k + i] = od.P[lg + k][i];
			// Q[i] = (A^L)^(i+1) in extended precision
			long double A[4] = { -(long double) od.c[3], 1.0L, -(long double) od.c[4], 0.0L }, AL[4] = { 1.0L, 0.0L, 0.0L, 1.0L }, t[4];
			for (int i = 0; i < CASCADE_L; ++i) { mat_mul(AL, A, t); memcpy(AL, t, sizeof(t)); }
			long double Q[4] = { AL[0], AL[1], AL[2], AL[3] };
			for (int i = 0; i < 16; ++i) {
				for (int k = 0; k < 4; ++k) q[e * FQ_DOUBLES + 4 * i + k] = (double) Q[k];
				mat_mul(Q, AL, t);
				memcpy(Q, t, sizeof(t));
			}
		}
		if (!fops.upload(tab.data(), tab.size() * sizeof(double))) return false;
		if (!fq.upload(q.data(), q.size() * sizeof(double))) return false;
	}
	// tables of cascade_rows: one set of wave-uniform constants per channel PAIR -- only when the two channels of every pair
	// run identical ops (the usual case: one filter bank on all channels of a stream) and those ops are biquad sections and
	// gains.  A gain in front of a section is folded into that section's b coefficients (same states, the product rounds
	// differently in the last bit -- the sections are not bit-exact anyway), gains behind the last section become one factor
	// applied to the finished tile ([1] of the last op's entry); `add`, unselected channels and chains without any section
	// stay with cascade_fast (a pure gain chain must remain bit-exact).
	rows4_ok = false;
	if (ch_in % 2 == 0) {
		bool uniform = true, any_biquad = false, quad = (ch_in % 4 == 0);
		auto same = [](const OpDesc &a, const OpDesc &b) { return a.kind == b.kind && a.g == b.g && memcmp(a.c, b.c, sizeof(a.c)) == 0; };
		for (int c = 0; c < ch_in && uniform; ++c)
			for (int j = 0; j < n_ops && uniform; ++j) {
				const OpDesc &a = host[(size_t) (c & ~1) * n_ops + j], &b = host[(size_t) c * n_ops + j];
				if (!same(a, b)) uniform = false;
				if (a.kind != OP_BIQUAD && a.kind != OP_MUL) uniform = false;
				if (a.kind == OP_BIQUAD) any_biquad = true;
				if (quad && !same(host[(size_t) (c & ~3) * n_ops + j], b)) quad = false;
			}
		if (uniform && any_biquad) {
			const int n_pairs = ch_in / 2;
			std::vector<double> tab((size_t) n_pairs * n_ops * FOP_DOUBLES, 0.0), q((size_t) n_pairs * n_ops * FQ_DOUBLES, 0.0);
			int lg = 0;
			while ((1 << lg) < ROWS_L) ++lg;
			for (int g = 0; g < n_pairs; ++g) {
				double gain = 1.0;
				for (int j = 0; j < n_ops; ++j) {
					const OpDesc &od = host[(size_t) (2 * g) * n_ops + j];
					double *d = &tab[((size_t) g * n_ops + j) * FOP_DOUBLES];
					long long kind = (od.kind == OP_BIQUAD) ? OP_BIQUAD : OP_SKIP;
					memcpy(&d[0], &kind, sizeof(kind));
					d[1] = 1.0;
					if (od.kind == OP_MUL) { gain *= od.g; continue; }
					for (int i = 0; i < 5; ++i) d[2 + i] = od.c[i];
					for (int i = 0; i < 3; ++i) d[2 + i] *= gain;            // y = H(g x): b coefficients scaled, a and the states untouched
					gain = 1.0;
					for (int k = 0; k < 4; ++k) for (int i = 0; i < 4; ++i) d[FOP_PW + 4 * k + i] = od.P[lg + k][i];
					// Q[i] = (A^L)^(i+1), L = ROWS_L, in extended precision
					long double A[4] = { -(long double) od.c[3], 1.0L, -(long double) od.c[4], 0.0L }, AL[4] = { 1.0L, 0.0L, 0.0L, 1.0L }, t[4];
					for (int i = 0; i < ROWS_L; ++i) { mat_mul(AL, A, t); memcpy(AL, t, sizeof(t)); }
					long double Q[4] = { AL[0], AL[1], AL[2], AL[3] };
					for (int i = 0; i < 16; ++i) {
						for (int k = 0; k < 4; ++k) q[((size_t) g * n_ops + j) * FQ_DOUBLES + 4 * i + k] = (double) Q[k];
						mat_mul(Q, AL, t);
						memcpy(Q, t, sizeof(t));
					}
				}
				tab[((size_t) g * n_ops + (n_ops - 1)) * FOP_DOUBLES + 1] = gain;   // gains behind the last section
			}
			if (!frows.upload(tab.data(), tab.size() * sizeof(double))) return false;
			if (!frq.upload(q.data(), q.size() * sizeof(double))) return false;
			rows4_ok = quad;
		}
	}
	if (!state.alloc((size_t) S * ch_in * n_ops * 2 * sizeof(double))) return false;
	// channel group per workgroup: the whole stream when it fits in LDS (contiguous, vectorisable loads)
	Cg = (ch_in <= 16) ? ch_in : 8;
	while (Cg > 1 && cascade_lds_bytes(Cg, n_ops) > 150 * 1024) Cg = (Cg + 1) / 2;
	return true;
}

std::string CascadeStage::describe() const
{
	std::ostringstream o;
	o << "cascade[";
	for (size_t i = 0; i < names.size(); ++i) o << (i ? " " : "") << names[i];
	o << "; Cg=" << Cg << (ring.base ? (write_interleaved ? "; +ring" : "; ->ring") : "") << "]";
	return o.str();
}

// K chunks of len frames each, or false: not worth it / not possible.  The chunks must be whole tiles of whichever kernel
// launch_cascade will pick for S K C channels (cascade_rows<4>: 512 frames, <2>: 1024, <1>: 2048; cascade_fast: 1024).
bool CascadeStage::choose_chunks(long frames, int *K_out, long *len_out) const
{
	const char *env_s = getenv("DSP_AMD_CASCADE_CHUNKS");        // 0 = never, K = force up to K chunks (read per plan: tests switch it)
	const int env = env_s ? atoi(env_s) : -1;
	const long channels = (long) S * ch_in;
	const int D = 2 * n_ops;
	if (env == 0 || !chunk_linear || n_ops < 1 || D > 64) return false;
	if (env < 0) {
		if (channels > 512 || frames < 8192) return false;
		// Three launches against one: estimated times in microseconds, fitted to scripts/exp_chunk_threshold.sh.  Direct: the
		// P <= 8 skewed waves of a channel finish a 2048-frame tile every (n_ops + 1) x 1.9 us, or the chip is full (1.17e6 section-samples
		// per us for one channel per wave); chunked: cascade_rows<4> at full rate, the scan, one more pass over the output.
		const double work = (double) channels * frames * n_ops;
		const double t_direct = 20.0 + std::max(frames * (n_ops + 1) * 1.9 / (2048.0 * 8.0), work / 1.17e6);
		const double t_chunk = std::max(16.0, 1.25 * work / 2.06e6) + 8.0 + std::max(16.0, (double) channels * frames * 16.0 / 4.0e6);
		if (t_chunk > 0.8 * t_direct) return false;
	}
	const long k_target = (env > 0) ? env : (2048 + channels - 1) / channels;
	for (long unit : { 512L, 1024L, 2048L }) {
		if (frames % unit) continue;
		const long f = frames / unit;
		long K = 1;
		auto scan_fits = [&](long k) {                      // cascade_chunk_carry: K D doubles of LDS, (groups x D) threads
			long g = 1;
			while (g * g < k) ++g;
			return k * D <= 8192 && ((k + g - 1) / g) * D <= 1024;
		};
		for (long k = std::min(f, k_target); k >= 1; --k) if (f % k == 0 && scan_fits(k)) { K = k; break; }
		if (K < 4) continue;
		const long len = frames / K, V = channels * K;
		const long tile = (V >= 1024 && rows4_ok) ? 512 : frows.p ? (V >= 512 ? 1024 : 2048) : 1024;
		if (len % tile) continue;
		if ((double) len * D * sizeof(double) * ch_in > 256e6) continue;
		*K_out = (int) K; *len_out = len;
		return true;
	}
	return false;
}

// M (state after len frames of silence, per unit state) and H (the outputs on the way) by running the sections themselves,
// in extended precision; channels with identical ops share one table
CascadeStage::ChunkPlan::~ChunkPlan() { if (done) (void) hipEventDestroy(done); }

CascadeStage::ChunkPlan *CascadeStage::chunk_plan_for(long frames, int K, long len)
{
	// plans that fell out of the cache are released once the launches that used them have finished -- asked with an event
	// query, never waited for: a new call size must not stall the device or break the caller's asynchronous stream
	for (size_t i = 0; i < retired_plans.size();) {
		// (a plan whose event could not be created: nothing to ask -- wait for the device before its buffers go)
		if (!retired_plans[i]->done) { (void) hipDeviceSynchronize(); retired_plans.erase(retired_plans.begin() + i); }
		else if (hipEventQuery(retired_plans[i]->done) == hipSuccess) retired_plans.erase(retired_plans.begin() + i);
		else { (void) hipGetLastError(); ++i; }
	}
	for (size_t i = 0; i < chunk_plans.size(); ++i)
		if (chunk_plans[i]->frames == frames && chunk_plans[i]->K == K && chunk_plans[i]->len == len) {
			std::rotate(chunk_plans.begin(), chunk_plans.begin() + i, chunk_plans.begin() + i + 1);
			return chunk_plans.front().get();
		}
	std::unique_ptr<ChunkPlan> c(new ChunkPlan);
	if (!build_chunk_plan(*c, frames, K, len)) return nullptr;
	if (hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); c->done = nullptr; }
	if (chunk_plans.size() >= 8) {
		retired_plans.push_back(std::move(chunk_plans.back()));
		chunk_plans.pop_back();
		if (retired_plans.size() > 16) {                  // a caller that never lets the stream finish: wait for the oldest one only
			if (retired_plans.front()->done) (void) hipEventSynchronize(retired_plans.front()->done);
			retired_plans.erase(retired_plans.begin());
		}
	}
	chunk_plans.insert(chunk_plans.begin(), std::move(c));
	return chunk_plans.front().get();
}

bool CascadeStage::build_chunk_plan(ChunkPlan &chunk, long frames, int K, long len)
{
	const int D = 2 * n_ops;
	std::vector<int> cls(ch_in, -1);
	std::vector<int> rep;
	auto same = [&](int a, int b) {
		for (int j = 0; j < n_ops; ++j) {
			const OpDesc &x = host_ops[(size_t) a * n_ops + j], &y = host_ops[(size_t) b * n_ops + j];
			if (x.kind != y.kind || x.g != y.g || memcmp(x.c, y.c, sizeof(x.c)) != 0) return false;
		}
		return true;
	};
	for (int c = 0; c < ch_in; ++c) {
		for (size_t r = 0; r < rep.size() && cls[c] < 0; ++r) if (same(rep[r], c)) cls[c] = (int) r;
		if (cls[c] < 0) { cls[c] = (int) rep.size(); rep.push_back(c); }
	}
	int n_pow = 1;                                         // scan group size g = ceil(sqrt(K)): tables M, M^2 .. M^g
	while (n_pow * n_pow < K) ++n_pow;
	std::vector<double> H(rep.size() * (size_t) len * D, 0.0), Mp(rep.size() * (size_t) n_pow * D * D, 0.0);
	std::vector<long double> M((size_t) D * D), T((size_t) D * D), st(D);
	for (size_t r = 0; r < rep.size(); ++r) {
		const OpDesc *od = &host_ops[(size_t) rep[r] * n_ops];
		std::fill(M.begin(), M.end(), 0.0L);
		for (int k = 0; k < D; ++k) {
			if (od[k / 2].kind != OP_BIQUAD) continue;          // gains carry no state: zero column
			std::fill(st.begin(), st.end(), 0.0L);
			st[k] = 1.0L;
			for (long i = 0; i < len; ++i) {
				long double v = 0.0L;
				for (int j = 0; j < n_ops; ++j) {
					if (od[j].kind == OP_MUL) { v *= (long double) od[j].g; continue; }
					if (od[j].kind != OP_BIQUAD) continue;
					const long double y = (long double) od[j].c[0] * v + st[2 * j];                                  // biquad.h:76-92
					st[2 * j] = (long double) od[j].c[1] * v - (long double) od[j].c[3] * y + st[2 * j + 1];
					st[2 * j + 1] = (long double) od[j].c[2] * v - (long double) od[j].c[4] * y;
					v = y;
				}
				H[(r * (size_t) len + i) * D + k] = (double) v;
			}
			for (int q = 0; q < D; ++q) M[(size_t) q * D + k] = st[q];
		}
		std::vector<long double> Pw = M;                       // M^(pw + 1)
		for (int pw = 0; pw < n_pow; ++pw) {
			for (size_t e = 0; e < Pw.size(); ++e) Mp[((r * (size_t) n_pow + pw) * D * D) + e] = (double) Pw[e];
			for (int a = 0; a < D; ++a)
				for (int b = 0; b < D; ++b) {
					long double acc = 0.0L;
					for (int k = 0; k < D; ++k) acc += Pw[(size_t) a * D + k] * M[(size_t) k * D + b];
					T[(size_t) a * D + b] = acc;
				}
			Pw = T;
		}
	}
	chunk.frames = 0;
	if (!chunk.cls.upload(cls.data(), cls.size() * sizeof(int)) || !chunk.H.upload(H.data(), H.size() * sizeof(double)) ||
	    !chunk.Mp.upload(Mp.data(), Mp.size() * sizeof(double)) ||
	    !chunk.cstate.alloc((size_t) S * K * ch_in * D * sizeof(double)) || !chunk.X.alloc((size_t) S * K * ch_in * D * sizeof(double))) return false;
	chunk.frames = frames; chunk.len = len; chunk.K = K; chunk.n_pow = n_pow; chunk.n_cls = (int) rep.size();
	return true;
}

CascadeParams CascadeStage::params(const double *in, long in_stride, ssize_t frames, double *out, long out_stride) const
{
	CascadeParams p;
	p.in = in; p.out = out;
	p.in_fmt = wire_in_fmt;
	p.sink = wire_sink;
	p.in_stride_frames = in_stride; p.out_stride_frames = out_stride;
	p.frames = frames;
	p.C = ch_in; p.cg0 = 0; p.Cg = Cg;
	p.n_ops = n_ops;
	p.ops = ops.as<OpDesc>();
	p.fops = fops.as<double>();
	p.fq = fq.as<double>();
	p.frows = frows.p ? frows.as<double>() : nullptr;
	p.frq = frq.p ? frq.as<double>() : nullptr;
	p.rows4_ok = rows4_ok ? 1 : 0;
	p.state = state.as<double>();
	p.ring = ring;
	p.write_interleaved = write_interleaved;
	p.xcd_map = 1;
	return p;
}

// the wire formats are spoken by cascade_rows (+ the generic kernel on what is left of the block behind its tiles)
bool CascadeStage::wire_ok(int in_fmt, bool sink_on, int out_fmt, const void *in, long in_stride, const void *out, long out_stride, ssize_t frames) const
{
	if (!wire_fusion_on()) return false;
	int K = 0;
	long len = 0;
	if (!ring.base && write_interleaved && (S == 1 || (in_stride == frames && out_stride == frames)) && choose_chunks(frames, &K, &len)) return false;
	CascadeParams p = params(static_cast<const double *>(in), in_stride, frames, static_cast<double *>(const_cast<void *>(out)), out_stride);
	p.in_fmt = in_fmt;
	p.sink.on = sink_on ? 1 : 0;
	p.sink.fmt = out_fmt;
	return cascade_rows_takes(p, S);
}

bool CascadeStage::wire_in_ok(int fmt, const void *in, long in_stride, ssize_t frames, bool also_out, int out_fmt) const
{
	// (a call the fused first pass takes: its kernels read the format themselves -- what run() will decide by asking the same question)
	if (wire_fusion_on() && !also_out && fuse_probe && ring.base && !write_interleaved && fuse_probe(in, in_stride, frames, fmt)) return true;
	// (the destination of a first-but-not-last stage is one of the pipeline's own aligned buffers, or the ring)
	return wire_ok(fmt, also_out, out_fmt, in, in_stride, nullptr, in_stride, frames);
}

bool CascadeStage::wire_out_ok(int fmt, const void *out, long out_stride, ssize_t frames, bool also_in, int in_fmt) const
{
	return wire_ok(also_in ? in_fmt : PCM_DOUBLE, true, fmt, nullptr, out_stride, out, out_stride, frames);
}

const CascadeStage::FuseTables &CascadeStage::fuse_tables()
{
	FuseTables &ft = fuse_tab;
	if (ft.tried) return ft;
	ft.tried = true;
	if (ch_in < 1 || n_ops < 1) return ft;
	auto same = [](const OpDesc &a, const OpDesc &b) { return a.kind == b.kind && a.g == b.g && memcmp(a.c, b.c, sizeof(a.c)) == 0; };
	// every channel alike (one table), or at least the two channels of every pair (one table per pair)
	bool uniform = true, pairwise = (ch_in % 2) == 0;
	for (int j = 0; j < n_ops; ++j)
		for (int c = 1; c < ch_in; ++c) {
			if (!same(host_ops[(size_t) c * n_ops + j], host_ops[j])) uniform = false;
			if ((c & 1) && !same(host_ops[(size_t) c * n_ops + j], host_ops[(size_t) (c - 1) * n_ops + j])) pairwise = false;
		}
	if (!uniform && !pairwise) return ft;
	// section slots: the ops that are a section in any channel (a selector makes them OP_SKIP in the others); gains have no slot
	std::vector<int> sec_op;
	for (int j = 0; j < n_ops; ++j) {
		bool section = false, gain_here = false;
		for (int c = 0; c < ch_in; ++c) {
			const int k = host_ops[(size_t) c * n_ops + j].kind;
			if (k == OP_BIQUAD) section = true;
			else if (k == OP_MUL) gain_here = true;
			else if (k != OP_SKIP) return ft;                         // (`add` is not linear in the state)
		}
		// an op that is a section on some channels and a gain on others has no slot form (the table loop below would come out an entry short
		// for the pairs it is a gain for): such a chain keeps the separate kernels
		if (section && gain_here) return ft;
		if (section) sec_op.push_back(j);
	}
	const int slots = fused_section_slots((int) sec_op.size());
	if (sec_op.empty() || !slots) return ft;
	ft.n_real = (int) sec_op.size();
	const int tables = uniform ? 1 : ch_in / 2;
	std::vector<double> sec, gains;
	for (int t = 0; t < tables; ++t) {
		const OpDesc *od = &host_ops[(size_t) (2 * t) * n_ops];
		double gain = 1.0;
		size_t k = 0;
		for (int j = 0; j < n_ops; ++j) {
			if (od[j].kind == OP_MUL) { gain *= od[j].g; continue; }
			if (k >= sec_op.size() || sec_op[k] != j) continue;      // (an op that is OP_SKIP in every channel)
			// y = H(g x): the b coefficients scaled, states untouched; a pair the section is not for passes g x through (states stay zero)
			if (od[j].kind == OP_BIQUAD) sec.insert(sec.end(), { od[j].c[0] * gain, od[j].c[1] * gain, od[j].c[2] * gain, od[j].c[3], od[j].c[4], 0.0 });
			else sec.insert(sec.end(), { gain, 0.0, 0.0, 0.0, 0.0, 0.0 });
			gain = 1.0;
			++k;
		}
		for (int pad = (int) sec_op.size(); pad < slots; ++pad) sec.insert(sec.end(), { 1.0, 0.0, 0.0, 0.0, 0.0, 0.0 });   // pass-through
		gains.push_back(gain);
	}
	while ((int) sec_op.size() < slots) sec_op.push_back(-1);
	if (sec.size() != (size_t) tables * slots * 6) return ft;         // (what the fused kernels index with sec_stride)
	if (!ft.sec.upload(sec.data(), sec.size() * sizeof(double)) || !ft.sec_op.upload(sec_op.data(), sec_op.size() * sizeof(int))) return ft;
	if (tables > 1 && !ft.gain_tab.upload(gains.data(), gains.size() * sizeof(double))) return ft;
	ft.n_sec = slots;
	ft.pairs = tables;
	ft.gain = gains[0];
	ft.ok = true;
	return ft;
}

// Gt[t][d], d = 2 kk + b over the chain's sections kk (channel 0: the fused path is for chains whose channels agree): the state (m0, m1)[b]
// of section kk that a unit sample at frame t of a zero-state chunk of plan.len frames leaves behind at the chunk's end -- the sections
// themselves run on an impulse in extended precision, read backwards
bool CascadeStage::fuse_gtable(ChunkPlan &plan)
{
	if (plan.G.p) return plan.g_states > 0;
	if (fuse_tables().pairs != 1) return false;          // (one table for every column of the product: chains whose channels all agree)
	std::vector<int> secs;
	for (int j = 0; j < n_ops; ++j) if (host_ops[j].kind == OP_BIQUAD) secs.push_back(j);
	if (secs.empty() || secs.size() > 16) return false;
	const long len = plan.len;
	std::vector<double> G((size_t) len * 32, 0.0);
	std::vector<long double> st((size_t) 2 * n_ops, 0.0L);
	for (long n = 0; n < len; ++n) {
		long double v = (n == 0) ? 1.0L : 0.0L;
		for (int j = 0; j < n_ops; ++j) {
			const OpDesc &od = host_ops[j];
			if (od.kind == OP_MUL) { v *= (long double) od.g; continue; }
			if (od.kind != OP_BIQUAD) continue;
			const long double y = (long double) od.c[0] * v + st[2 * j];                                  // biquad.h:76-92
			st[2 * j] = (long double) od.c[1] * v - (long double) od.c[3] * y + st[2 * j + 1];
			st[2 * j + 1] = (long double) od.c[2] * v - (long double) od.c[4] * y;
			v = y;
		}
		double *row = &G[(size_t) (len - 1 - n) * 32];
		for (size_t kk = 0; kk < secs.size(); ++kk) { row[2 * kk] = (double) st[2 * secs[kk]]; row[2 * kk + 1] = (double) st[2 * secs[kk] + 1]; }
	}
	if (!plan.G.upload(G.data(), G.size() * sizeof(double))) return false;
	plan.g_states = (int) (2 * secs.size());
	return true;
}

ssize_t CascadeStage::run(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st)
{
	CascadeParams p = params(in, in_stride, frames, out, out_stride);
	int K = 0;
	long len = 0;
	const bool wire = wire_in_fmt != PCM_DOUBLE || wire_sink.on;
	pending = Pending();
	if (fuse_probe && !wire_sink.on && ring.base && !write_interleaved && fuse_probe(in, in_stride, frames, wire_in_fmt)) {
		// the convolver behind takes the call whole (ConvStage::run_fused; a wire format is read by its kernels' own loads): nothing to launch here
		pending.in = in; pending.in_stride = in_stride; pending.frames = frames; pending.in_fmt = wire_in_fmt;
		return frames;
	}
	if (!wire && !ring.base && write_interleaved && (S == 1 || (in_stride == frames && out_stride == frames)) && choose_chunks(frames, &K, &len)) {
		ChunkPlan *cpl = chunk_plan_for(frames, K, len);
		if (!cpl) return -1;
		ChunkPlan &chunk = *cpl;
		// S K zero-state "streams" of len frames, then the carried states and the correction
		p.frames = chunk.len;
		p.in_stride_frames = p.out_stride_frames = chunk.len;
		p.state = chunk.cstate.as<double>();
		(void) hipMemsetAsync(chunk.cstate.p, 0, chunk.cstate.bytes, st);
		{ ProfScope ps("cascade_kernel", st); ps.rename(launch_cascade(p, S * chunk.K, st)); }
		ChunkParams cp;
		cp.out = out; cp.out_stride_frames = out_stride; cp.len = chunk.len;
		cp.C = ch_in; cp.K = chunk.K; cp.D = 2 * n_ops; cp.n_pow = chunk.n_pow; cp.n_cls = chunk.n_cls;
		cp.cls = chunk.cls.as<int>(); cp.H = chunk.H.as<double>(); cp.Mp = chunk.Mp.as<double>();
		cp.cstate = chunk.cstate.as<double>(); cp.X = chunk.X.as<double>(); cp.state = state.as<double>();
		{ ProfScope ps("cascade_chunk_carry", st); launch_chunk_carry(cp, S, st); }
		{ ProfScope ps("cascade_chunk_fix", st); launch_chunk_fix(cp, S, st); }
		if (chunk.done) (void) hipEventRecord(chunk.done, st);
		return frames;
	}
	{ ProfScope ps("cascade_kernel", st); ps.rename(launch_cascade(p, S, st)); }
	if (ring.base) ring.pos = (ring.pos + frames) & ring.mask;
	return frames;
}

void CascadeStage::reset(hipStream_t st)
{
	(void) hipMemsetAsync(state.p, 0, state.bytes, st);
	pending = Pending();
}

// -------------------------------------------------------------- RemixStage

bool RemixStage::init(const Spec &sp)
{
	max_n = 1;
	if (sp.kind == Kind::Mix) {
		weighted = true;
		for (auto &r : sp.mix_idx) max_n = std::max<int>(max_n, (int) r.size());
		std::vector<int> idx((size_t) ch_out * max_n, -1);
		std::vector<double> w((size_t) ch_out * max_n, 0.0);
		for (int k = 0; k < ch_out; ++k)
			for (size_t j = 0; j < sp.mix_idx[k].size(); ++j) { idx[(size_t) k * max_n + j] = sp.mix_idx[k][j]; w[(size_t) k * max_n + j] = sp.mix_w[k][j]; }
		if (!sp.mix_post.empty() && !d_post.upload(sp.mix_post.data(), sp.mix_post.size() * sizeof(double))) return false;
		return d_idx.upload(idx.data(), idx.size() * sizeof(int)) && d_w.upload(w.data(), w.size() * sizeof(double));
	}
	for (auto &r : sp.remix) max_n = std::max(max_n, num_set(r));
	std::vector<int> idx((size_t) ch_out * max_n, -1);
	for (int k = 0; k < ch_out; ++k) {
		int n = 0;
		for (int j = 0; j < ch_in; ++j) if (sp.remix[k][j]) idx[(size_t) k * max_n + n++] = j;
	}
	return d_idx.upload(idx.data(), idx.size() * sizeof(int));
}

ssize_t RemixStage::run(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st)
{
	RemixParams p{ in, out, in_stride, out_stride, frames, ch_in, ch_out, d_idx.as<int>(), max_n,
	               weighted ? d_w.as<double>() : nullptr, (weighted && d_post.p) ? d_post.as<double>() : nullptr, wire_in_fmt, wire_sink };
	{ ProfScope ps("remix_kernel", st); launch_remix(p, S, st); }
	return frames;
}

// -------------------------------------------------------------- DelayStage

bool DelayStage::init(const Spec &sp)
{
	len = sp.delay;
	discard = sp.discard;
	remaining_discard = discard;
	std::vector<long> l(ch_in), off(ch_in);
	ring_per_stream = 0;
	max_len = 0;
	for (int k = 0; k < ch_in; ++k) {
		l[k] = (long) len[k];
		off[k] = ring_per_stream;
		ring_per_stream += l[k];
		max_len = std::max(max_len, l[k]);
	}
	if (!d_len.upload(l.data(), l.size() * sizeof(long))) return false;
	if (!d_off.upload(off.data(), off.size() * sizeof(long))) return false;
	// two copies (read / write) per stream, see delay_kernel
	return ring.alloc((size_t) 2 * S * std::max<long>(ring_per_stream, 1) * sizeof(double));
}

std::string DelayStage::describe() const
{
	std::ostringstream o;
	o << "align[max_len=" << max_len << " discard=" << discard << "]";
	return o.str();
}

void launch_delay_ex(const DelayParams &p, long ring_alt_off, long skip, long max_len, int n_streams, hipStream_t stream);

ssize_t DelayStage::run(const double *in, long in_stride, ssize_t frames, double *out, long out_stride, hipStream_t st)
{
	const long half = (long) S * std::max<long>(ring_per_stream, 1);
	DelayParams p;
	p.in = in; p.out = out;
	p.in_stride_frames = in_stride; p.out_stride_frames = out_stride;
	p.frames = frames;
	p.C = ch_in;
	p.len = d_len.as<long>();
	p.ring_off = d_off.as<long>();
	p.ring = ring.as<double>() + (size_t) (phase ? half : 0);
	p.ring_per_stream = ring_per_stream;
	p.pos = pos;
	p.in_fmt = wire_in_fmt;
	p.sink = wire_sink;
	const long skip = std::min<long>(remaining_discard, frames);
	{ ProfScope ps("delay_kernel", st); launch_delay_ex(p, phase ? -half : half, skip, max_len, S, st); }
	phase ^= 1;
	pos += frames;
	remaining_discard -= skip;
	return frames - skip;
}

void DelayStage::reset(hipStream_t st)
{
	(void) hipMemsetAsync(ring.p, 0, ring.bytes, st);
	pos = 0;
	phase = 0;
	remaining_discard = discard;
}

// ---------------------------------------------------------------- Pipeline

Pipeline::~Pipeline() {}

std::unique_ptr<Pipeline> Pipeline::compile(const std::vector<const Spec *> &specs, int fs, int channels, int n_streams, ssize_t max_frames)
{
	std::unique_ptr<Pipeline> pl(new Pipeline);
	pl->S = n_streams;
	pl->fs_in = fs;
	pl->ch_in = channels;
	pl->max_frames = max_frames;
	int cur_fs = fs, cur_ch = channels;
	CascadeStage *casc = nullptr;
	auto flush = [&]() -> bool {
		if (!casc) return true;
		const bool ok = casc->finalize();
		casc = nullptr;
		return ok;
	};
	auto base = [&](Stage *s, const Spec &sp) {
		s->S = n_streams;
		s->fs_in = sp.fs_in; s->fs_out = sp.fs_out;
		s->ch_in = sp.ch_in; s->ch_out = sp.ch_out;
	};
	ssize_t frames_here = max_frames;   // worst-case frames entering the next stage
	std::vector<std::unique_ptr<Spec>> merged;   // specs synthesised here (LTI merges)
	for (size_t si = 0; si < specs.size(); ++si) {
		const Spec *sp = specs[si];
		// OPT-IN (DSP_AMD_MERGE_IIR=1, off by default): biquad sections and gains on every channel directly in front of a zero-latency
		// convolution are LTI stages like the two cases below, and once their joint impulse response has decayed below 2^-70 of
		// its peak it IS a finite filter as far as fp64 can tell: g * h (T + L - 1 taps) through the convolver alone, no cascade
		// pass at all.  Exactness is the decay criterion's (checked per chain: a 20 Hz high-pass needs 26000 taps, a 2 Hz one
		// does not merge); the price is L frames less hop per transform.  Off by default because the headline workload is
		// DEFINED as a biquad cascade + fir_p and is measured as such; bench.py --chain runs report the merged plan as a side figure.
		if ((sp->kind == Kind::Biquad || sp->kind == Kind::Gain) && !casc && getenv("DSP_AMD_MERGE_IIR") && atoi(getenv("DSP_AMD_MERGE_IIR")) > 0) {
			auto uniform = [](const Spec *q) {
				if (num_set(q->sel) != q->ch_in || q->ch_in != q->ch_out || q->frac_delay || !q->bq_more.empty() || !q->riir.empty()) return false;
				if (q->kind == Kind::Biquad) { for (int c = 1; c < q->ch_in; ++c) if (q->bq[c] != q->bq[0]) return false; return true; }
				if (q->kind == Kind::Gain) { for (int c = 1; c < q->ch_in; ++c) if (q->vec[c] != q->vec[0]) return false; return true; }
				return false;
			};
			size_t k = si;
			while (k < specs.size() && uniform(specs[k])) ++k;
			const Spec *cv = (k > si && k < specs.size()) ? specs[k] : nullptr;
			if (cv && cv->kind == Kind::Conv && cv->conv_mode == CONV_ZERO_LATENCY && cv->latency == 0 && cv->fch == 1 && num_set(cv->sel) == cv->ch_in
			    && cv->ch_in == sp->ch_in && cv->riir.empty()) {
				// joint impulse response of the sections, in extended precision, until a 256-sample block stays below 2^-70 of the peak
				const long Lmax = std::max<long>(cv->T, 1L << 16);
				std::vector<long double> g;
				std::vector<std::array<long double, 2>> st(k - si, std::array<long double, 2>{ { 0.0L, 0.0L } });
				long double peak = 0.0L, blk = 0.0L;
				bool decayed = false;
				for (long n = 0; n < Lmax && !decayed; ++n) {
					long double v = (n == 0) ? 1.0L : 0.0L;
					for (size_t q = si; q < k; ++q) {
						const Spec *b = specs[q];
						if (b->kind == Kind::Gain) { v *= (long double) b->vec[0]; continue; }
						const std::array<double, 5> &c = b->bq[0];             // TDF-II, biquad.h:76-92
						std::array<long double, 2> &m = st[q - si];
						const long double r = (long double) c[0] * v + m[0];
						m[0] = (long double) c[1] * v + m[1] - (long double) c[3] * r;
						m[1] = (long double) c[2] * v - (long double) c[4] * r;
						v = r;
					}
					g.push_back(v);
					peak = std::max(peak, fabsl(v));
					blk = std::max(blk, fabsl(v));
					if ((n & 255) == 255) { if (n >= 511 && blk < ldexpl(peak, -70)) decayed = true; blk = 0.0L; }
				}
				if (decayed) {
					const long L = (long) g.size();
					merged.emplace_back(new Spec(*cv));
					Spec &m = *merged.back();
					m.T = cv->T + L - 1;
					m.taps.assign((size_t) m.T, 0.0);
					std::vector<double> gd(g.begin(), g.end());
					for (ssize_t i = 0; i < cv->T; ++i) {
						const double a = cv->taps[i];
						if (a == 0.0) continue;
						double *dst = &m.taps[i];
						for (long j = 0; j < L; ++j) dst[j] += a * gd[j];
					}
					m.name.clear();
					for (size_t q = si; q < k; ++q) m.name += specs[q]->name + "+";
					m.name += cv->name;
					log_msg(LL_VERBOSE, "info: %zu sections and gains folded into %s: %ld taps of joint impulse response, %zd taps in all", k - si, cv->name.c_str(), L, m.T);
					sp = &m;
					si = k;
				}
			}
		}
		// fir_p -> fir_p (hilbert -p -> fir_p, ...): two zero-latency convolutions of every channel with one-channel
		// filters = ONE convolution with h1 * h2 (T1 + T2 - 1 taps; the chain's drain length is the same sum)
		while (sp->kind == Kind::Conv && si + 1 < specs.size() && specs[si + 1]->kind == Kind::Conv && !getenv("DSP_AMD_NO_LTI_MERGE")) {
			const Spec *b = specs[si + 1];
			const bool ok = sp->conv_mode == CONV_ZERO_LATENCY && b->conv_mode == CONV_ZERO_LATENCY && sp->latency == 0 && b->latency == 0
			                && sp->fch == 1 && b->fch == 1 && num_set(sp->sel) == sp->ch_in && num_set(b->sel) == b->ch_in && sp->ch_in == b->ch_in
			                && (double) sp->T * (double) b->T <= 2.0e9;
			if (!ok) break;
			{
				// only when the one convolution is cheaper than the two at this call size (transform sizes are powers of
				// two: a few taps more can push a call into a second, nearly empty block)
				double ca = 0, cb = 0, cm = 0;
				conv_plan(sp->T, frames_here, false, &ca);
				conv_plan(b->T, frames_here, false, &cb);
				conv_plan(sp->T + b->T - 1, frames_here, false, &cm);
				if (cm > ca + cb) break;
			}
			merged.emplace_back(new Spec(*sp));
			Spec &m = *merged.back();
			m.T = sp->T + b->T - 1;
			m.taps.assign((size_t) m.T, 0.0);
			for (ssize_t i = 0; i < sp->T; ++i) {
				const double a = sp->taps[i];
				if (a == 0.0) continue;
				double *dst = &m.taps[i];
				const double *hb = b->taps.data();
				for (ssize_t j = 0; j < b->T; ++j) dst[j] += a * hb[j];
			}
			m.name = sp->name + "+" + b->name;
			sp = &m;
			++si;
		}
		// fir_p -> integer-ratio resample: two LTI stages on all channels = ONE multi-phase convolution whose branch
		// filters are h * h_p (one set of FFT passes instead of two, no slab between them)
		if (sp->kind == Kind::Conv && si + 1 < specs.size() && specs[si + 1]->kind == Kind::Resample && !getenv("DSP_AMD_NO_LTI_MERGE")) {
			const Spec *rs = specs[si + 1];
			// (the same condition under which the resampler rides the FFT convolver, conv.cpp: a resampler on the polyphase kernels cannot take a filter in front)
			const bool int_ratio = (rs->rs_n == 1 || rs->rs_d == 1) && rs->rs_n <= 8 && rs->rs_d <= 8 && !getenv("DSP_AMD_RESAMPLE_DIRECT");
			bool cheaper = false;
			if (int_ratio) {
				// the branches of an n-fold upsampler cost about (1 + 2 n) / 3 of a plain convolution of the same size
				std::vector<double> tab; int J = 0; long od = 0;
				resample_polyphase_table(*rs, &J, &od, tab);
				const double w = (1.0 + 2.0 * rs->rs_n) / 3.0;
				double ca = 0, cb = 0, cm = 0;
				conv_plan(sp->T, frames_here, false, &ca);
				conv_plan(J, frames_here, true, &cb);
				conv_plan(sp->T + J - 1, frames_here, true, &cm);
				cheaper = (w * cm <= ca + w * cb);
			}
			if (int_ratio && cheaper && sp->conv_mode == CONV_ZERO_LATENCY && sp->latency == 0 && sp->fch == 1 && num_set(sp->sel) == sp->ch_in && sp->ch_in == rs->ch_in) {
				merged.emplace_back(new Spec(*rs));
				merged.back()->rs_pre = sp->taps;
				merged.back()->rs_pre_name = sp->name;
				sp = merged.back().get();
				++si;
			}
		}
		const Spec *parts[3] = { sp, nullptr, nullptr };
		int n_parts = 1;
		if (sp->kind == Kind::Crossfeed) {
			// crossfeed.c:33-50 as three device stages (see crossfeed_expand)
			Spec *q[3];
			for (int i = 0; i < 3; ++i) { merged.emplace_back(new Spec); q[i] = merged.back().get(); }
			crossfeed_expand(*sp, *q[0], *q[1], *q[2]);
			for (int i = 0; i < 3; ++i) parts[i] = q[i];
			n_parts = 3;
		}
		for (int pi = 0; pi < n_parts; ++pi) {
		sp = parts[pi];
		if (sp->fs_in != cur_fs || sp->ch_in != cur_ch) {
			set_error("pipeline: BUG: stream format mismatch at %s (%d ch @ %d vs %d ch @ %d)", sp->name.c_str(), sp->ch_in, sp->fs_in, cur_ch, cur_fs);
			return nullptr;
		}
		switch (sp->kind) {
		case Kind::Crossfeed: break;   // expanded above
		case Kind::Gain: case Kind::Add: case Kind::Biquad:
			if (!casc) {
				casc = new CascadeStage;
				base(casc, *sp);
				pl->stages.emplace_back(casc);
			}
			casc->add(*sp);
			for (size_t i = 0; i < sp->bq_more.size(); ++i) {           // the further sections of a high-order all-pass
				merged.emplace_back(new Spec(*sp));
				Spec &more = *merged.back();
				more.bq = sp->bq_more[i];
				more.sel = sp->sel_more[i];
				more.bq_more.clear(); more.sel_more.clear();
				more.name = sp->name + ":" + std::to_string(i + 2);
				casc->add(more);
			}
			break;
		case Kind::Delay:
			break;   // realised by Align (delay.c:142-147, 195-202)
		case Kind::Remix: case Kind::Mix: {
			if (!flush()) return nullptr;
			RemixStage *r = new RemixStage;
			base(r, *sp);
			pl->stages.emplace_back(r);
			if (!r->init(*sp)) return nullptr;
			break;
		}
		case Kind::Align: {
			if (!flush()) return nullptr;
			{
				bool only_discard = sp->discard > 0 && !pl->stages.empty();
				for (ssize_t len : sp->delay) if (len != 0) only_discard = false;
				if (only_discard && pl->stages.back()->absorb_discard((long) sp->discard)) break;
			}
			DelayStage *d = new DelayStage;
			base(d, *sp);
			pl->stages.emplace_back(d);
			if (!d->init(*sp)) return nullptr;
			break;
		}
		case Kind::FirDirect: case Kind::Conv: case Kind::Resample: {
			CascadeStage *feeder = casc;
			if (!flush()) return nullptr;
			Stage *prev = (!feeder && !pl->stages.empty()) ? pl->stages.back().get() : nullptr;
			Stage *s = make_conv_stage(*sp, n_streams, frames_here, feeder, prev);
			if (!s) return nullptr;
			base(s, *sp);
			pl->stages.emplace_back(s);
			break;
		}
		}
		cur_fs = sp->fs_out;
		cur_ch = sp->ch_out;
		if (!pl->stages.empty()) frames_here = pl->stages.back()->max_out_frames(frames_here);
		}
	}
	if (!flush()) return nullptr;
	pl->fs_out = cur_fs;
	pl->ch_out = cur_ch;
	// ping-pong scratch sized for the widest intermediate
	ssize_t fr = max_frames;
	size_t worst = 0;
	long worst_frames = max_frames;
	int worst_ch = channels;
	for (auto &s : pl->stages) {
		fr = s->max_out_frames(fr);
		worst_frames = std::max<long>(worst_frames, fr);
		worst_ch = std::max(worst_ch, s->ch_out);
	}
	worst = (size_t) n_streams * worst_frames * worst_ch * sizeof(double);
	if (pl->stages.size() > 1) {
		for (int i = 0; i < 2; ++i) {
			if (!pl->tmp[i].alloc(worst, false)) return nullptr;
			pl->tmp_stride[i] = worst_frames;
		}
	}
	pl->tmp_ch = worst_ch;
	return pl;
}

ssize_t Pipeline::max_out_frames(ssize_t in_frames) const
{
	ssize_t f = in_frames;
	for (auto &s : stages) f = s->max_out_frames(f);
	return f;
}

bool Pipeline::resident_plan(ResidentParams *rp, const int *fir_phase[]) const
{
	if (S != 1 || stages.empty() || stages.size() > (size_t) RES_MAX_PASSES) return false;
	memset(rp, 0, sizeof(*rp));
	rp->Cin = ch_in; rp->Cout = ch_out;
	int c = ch_in;
	for (size_t k = 0; k < stages.size(); ++k) {
		ResidentPass &ps = rp->pass[k];
		fir_phase[k] = nullptr;
		Stage *st = stages[k].get();
		if (RemixStage *rm = dynamic_cast<RemixStage *>(st)) {
			// (its tables -- weights, factors, sources -- have 8 KB of the wave's LDS)
			if (rm->ch_in != c || (size_t) rm->ch_out * rm->sources_per_row() * 12 + (size_t) rm->ch_out * 8 > (size_t) RES_TAB_DOUBLES * 8) return false;
			memset(&ps, 0, sizeof(ps));
			ps.kind = RES_PASS_REMIX; ps.c_in = rm->ch_in; ps.c_out = rm->ch_out;
			ps.idx = rm->device_idx(); ps.w = rm->device_w(); ps.post = rm->device_post(); ps.max_n = rm->sources_per_row();
		}
		else if (CascadeStage *cs = dynamic_cast<CascadeStage *>(st)) {
			if (rp->n_casc >= RES_MAX_CASCADES || cs->ch_in != c || cs->ring.base || !cs->write_interleaved || cs->n_ops < 1 || cs->n_ops > 16 || cs->ch_in > 32) return false;
			memset(&ps, 0, sizeof(ps));
			ps.kind = RES_PASS_CASCADE; ps.c_in = ps.c_out = cs->ch_in; ps.casc = rp->n_casc;
			rp->cs[rp->n_casc++] = ResidentParams::Casc{ cs->ch_in, cs->n_ops, cs->device_ops(), cs->device_state() };
		}
		else if (!fir_direct_view(st, &ps, &fir_phase[k]) || ps.c_in != c) return false;
		c = ps.c_out;
	}
	if (c != ch_out) return false;
	rp->n_pass = (int) stages.size();
	return true;
}

ssize_t Pipeline::run(const double *d_in, ssize_t frames, double *d_out, long out_stride, hipStream_t st, long in_stride)
{
	if (in_stride <= 0) in_stride = frames;
	if (in_stride < frames) { set_error("pipeline: input stride %ld shorter than the call (%zd frames)", in_stride, frames); return PIPE_FAILED; }
	if (frames > max_frames) { set_error("pipeline: %zd frames exceed max_frames=%zd", frames, max_frames); return PIPE_FAILED; }
	if (out_stride <= 0) out_stride = max_out_frames(frames);
	if (frames <= 0) return 0;
	if (stages.empty()) {
		launch_copy_slab(d_in, in_stride, d_out, out_stride, frames, 0, ch_in, S, st);
		return hip_ok(hipGetLastError(), "copy_slab") ? frames : -1;
	}
	const double *cur = d_in;
	long cur_stride = in_stride;
	ssize_t F = frames;
	int which = 0;
	for (size_t i = 0; i < stages.size(); ++i) {
		Stage *s = stages[i].get();
		const bool last = (i + 1 == stages.size());
		double *dst;
		long dst_stride;
		if (last) { dst = d_out; dst_stride = out_stride; }
		else if (s->in_place_ok() && cur != d_in) { dst = const_cast<double *>(cur); dst_stride = cur_stride; }
		else {
			dst = tmp[which].as<double>();
			// slabs are re-strided per stage: [S][tmp_frames][ch_out] always fits the allocation
			dst_stride = (long) (tmp[which].bytes / sizeof(double) / S / s->ch_out);
			which ^= 1;
		}
		F = s->run(cur, cur_stride, F, dst, dst_stride, st);
		if (F < 0) return F;
		// a launch that failed (bad configuration, LDS not granted on this device ...) must not pass stale memory on as audio
		if (!hip_ok(hipGetLastError(), s->type())) return PIPE_FAILED;
		cur = dst;
		cur_stride = dst_stride;
		if (F == 0) {
			// nothing came out (latency still filling): downstream stages see no frames this call
			return 0;
		}
	}
	return F;
}

ssize_t Pipeline::run_wire(int in_fmt, const void *d_in, long in_stride, ssize_t frames, const WireSink &sink_in, void *d_out, long out_stride, hipStream_t st, int *fused)
{
	if (fused) *fused = 0;
	if (!pcm_sample_bytes(in_fmt) || !pcm_sample_bytes(sink_in.fmt)) { set_error("pipeline: unknown wire format %d / %d", in_fmt, sink_in.fmt); return PIPE_FAILED; }
	const bool drain = (d_in == nullptr);
	if (frames <= 0) return drain ? -1 : 0;
	if (frames > max_frames) { if (drain) frames = max_frames; else { set_error("pipeline: %zd frames exceed max_frames=%zd", frames, max_frames); return PIPE_FAILED; } }
	if (in_stride <= 0) in_stride = frames;
	if (in_stride < frames) { set_error("pipeline: input stride %ld shorter than the call (%zd frames)", in_stride, frames); return PIPE_FAILED; }
	if (out_stride <= 0) out_stride = std::max<long>(1, max_out_frames(frames));
	WireSink sink = sink_in;
	sink.on = 1;
	// the stand-alone sink pass over a plain fp64 slab of this pipeline (what a last stage that cannot fuse leaves behind)
	auto own_out = [&](long *stride) -> double * {
		const long need = std::max<long>(1, max_out_frames(max_frames));
		if (!wire_tmp_out.p && !wire_tmp_out.alloc((size_t) S * need * ch_out * sizeof(double), false)) return nullptr;
		*stride = need;
		return wire_tmp_out.as<double>();
	};
	auto sink_pass = [&](const double *src, long src_stride, ssize_t F) -> bool {
		PcmWriteParams w;
		w.in = src; w.out = d_out;
		w.in_stride_frames = src_stride; w.out_stride_frames = out_stride; w.frames = F;
		w.C = ch_out; w.fmt = sink.fmt;
		w.dither_mult = sink.dither_mult; w.samples_before = sink.samples_before; w.stats = sink.stats;
		ProfScope ps("pcm_write", st);
		launch_pcm_write(w, S, st);
		return hip_ok(hipGetLastError(), "pcm_write");
	};
	if (drain) {
		// the rate changers' flush is a few frames at the end of a stream: plain drain2, then the sink pass
		long stride = 0;
		double *buf = own_out(&stride);
		if (!buf) return PIPE_FAILED;
		const ssize_t F = drain2(frames, buf, stride, st);
		if (F <= 0) return F;
		if (F > out_stride) { set_error("pipeline: %zd drained frames exceed the output stride %ld", F, out_stride); return PIPE_FAILED; }
		return sink_pass(buf, stride, F) ? F : PIPE_FAILED;
	}
	// ---- input side: the first stage's own loads, or read_buf_<fmt> as a pass of its own
	Stage *first = stages.empty() ? nullptr : stages.front().get();
	const bool single = stages.size() == 1;
	const double *cur = static_cast<const double *>(d_in);
	long cur_stride = in_stride;
	bool fin = false, fout_single = false;
	if (in_fmt != PCM_DOUBLE) {
		if (first && single && first->wire_in_ok(in_fmt, d_in, in_stride, frames, true, sink.fmt) && first->wire_out_ok(sink.fmt, d_out, out_stride, frames, true, in_fmt)) fin = fout_single = true;
		else if (first && !single && first->wire_in_ok(in_fmt, d_in, in_stride, frames, false, PCM_DOUBLE)) fin = true;
		else if (first && single && !first->wire_out_ok(sink.fmt, d_out, out_stride, frames, false, PCM_DOUBLE) && first->wire_in_ok(in_fmt, d_in, in_stride, frames, false, PCM_DOUBLE)) fin = true;
		if (!fin) {
			if (!wire_tmp_in.p && !wire_tmp_in.alloc((size_t) S * max_frames * ch_in * sizeof(double), false)) return PIPE_FAILED;
			PcmReadParams r{ d_in, wire_tmp_in.as<double>(), in_stride, (long) frames, (long) frames, ch_in, in_fmt };
			{ ProfScope ps("pcm_read", st); launch_pcm_read(r, S, st); }
			if (!hip_ok(hipGetLastError(), "pcm_read")) return PIPE_FAILED;
			cur = wire_tmp_in.as<double>();
			cur_stride = frames;
		}
	}
	if (stages.empty()) return sink_pass(cur, cur_stride, frames) ? frames : -1;
	ssize_t F = frames;
	int which = 0;
	int did = fin ? 1 : 0;
	for (size_t i = 0; i < stages.size(); ++i) {
		Stage *s = stages[i].get();
		const bool last = (i + 1 == stages.size());
		const bool wired_in = (i == 0 && fin);
		double *dst;
		long dst_stride;
		bool fout = false;
		if (last) {
			// decided with the frame count that actually reaches the last stage
			fout = single ? (fout_single || (!fin && s->wire_out_ok(sink.fmt, d_out, out_stride, F, false, PCM_DOUBLE)))
			              : s->wire_out_ok(sink.fmt, d_out, out_stride, F, false, PCM_DOUBLE);
			if (fout) { dst = static_cast<double *>(d_out); dst_stride = out_stride; }
			else { dst = own_out(&dst_stride); if (!dst) return PIPE_FAILED; }
		}
		else if (s->in_place_ok() && cur != d_in && !wired_in) { dst = const_cast<double *>(cur); dst_stride = cur_stride; }
		else {
			dst = tmp[which].as<double>();
			dst_stride = (long) (tmp[which].bytes / sizeof(double) / S / s->ch_out);
			which ^= 1;
		}
		if (wired_in) s->wire_in_fmt = in_fmt;
		if (fout) s->wire_sink = sink;
		F = s->run(cur, cur_stride, F, dst, dst_stride, st);
		s->wire_in_fmt = PCM_DOUBLE;
		s->wire_sink.on = 0;
		if (F < 0) return F;
		if (!hip_ok(hipGetLastError(), s->type())) return PIPE_FAILED;
		cur = dst;
		cur_stride = dst_stride;
		if (F == 0) return 0;
		if (last) {
			if (F > out_stride) { set_error("pipeline: %zd frames exceed the output stride %ld", F, out_stride); return PIPE_FAILED; }
			if (fout) did |= 2;
			else if (!sink_pass(cur, cur_stride, F)) return PIPE_FAILED;
		}
	}
	if (fused) *fused = did;
	return F;
}

ssize_t Pipeline::drain2(ssize_t block_frames, double *d_out, long out_stride, hipStream_t st)
{
	// mirror of effects_chain.c:1199-1217: give each stage with a drain2 a turn, feed what comes out to the rest
	while (drain_stage < (int) stages.size()) {
		Stage *s = stages[drain_stage].get();
		const bool last = (drain_stage + 1 == (int) stages.size());
		double *dst = last ? d_out : tmp[0].as<double>();
		long dst_stride = last ? out_stride : (long) (tmp[0].bytes / sizeof(double) / S / s->ch_out);
		ssize_t F = s->drain2(block_frames, dst, dst_stride, st);
		if (!hip_ok(hipGetLastError(), s->type()) || F == PIPE_FAILED) return PIPE_FAILED;
		if (F < 0) { ++drain_stage; continue; }
		const double *cur = dst;
		long cur_stride = dst_stride;
		int which = 1;
		for (size_t i = drain_stage + 1; i < stages.size() && F > 0; ++i) {
			Stage *n = stages[i].get();
			const bool nlast = (i + 1 == stages.size());
			double *d2 = nlast ? d_out : tmp[which].as<double>();
			long d2_stride = nlast ? out_stride : (long) (tmp[which].bytes / sizeof(double) / S / n->ch_out);
			which ^= 1;
			F = n->run(cur, cur_stride, F, d2, d2_stride, st);
			if (!hip_ok(hipGetLastError(), n->type()) || F < 0) return PIPE_FAILED;
			cur = d2;
			cur_stride = d2_stride;
		}
		return F;
	}
	return PIPE_DRY;
}

void Pipeline::reset(hipStream_t st)
{
	for (auto &s : stages) s->reset(st);
	drain_stage = 0;
}

std::string Pipeline::plan() const
{
	std::ostringstream o;
	o << "S=" << S << " " << ch_in << "ch@" << fs_in << " :";
	for (auto &s : stages) o << " " << s->describe();
	o << " : " << ch_out << "ch@" << fs_out;
	return o.str();
}

size_t Pipeline::device_bytes() const
{
	size_t b = tmp[0].bytes + tmp[1].bytes + wire_tmp_in.bytes + wire_tmp_out.bytes;
	for (auto &s : stages) b += s->device_bytes();
	return b;
}

}  // namespace dspamd
