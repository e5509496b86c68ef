// barprobe.hip -- round 6: where should the resident wave's doorbell and block live?
//
// Today (kernels_resident.hip, round 5) both are in page-locked HOST memory: the wave polls the doorbell over PCIe (a read round trip per poll), then reads
// the block over PCIe (a second, dependent round trip), writes the output and a completion word back (posted writes).  If device memory is host-writable
// (large BAR: the CPU stores straight into VRAM, posted), the wave can poll and read LOCAL memory and no PCIe read is left on the path.
//
// This probe answers, on the box it runs on:
//   1. can the CPU store into fine-grained device memory (hipExtMallocWithFlags / hipMalloc), and does a kernel see the stores?
//   2. the ping-pong time of one 8-byte doorbell: host writes seq -> a resident wave sees it -> writes seq to a host word -> host sees it,
//      with the doorbell (a) in host memory (today), (b) in device memory;
//   3. the same with a 1 KB block read by the wave behind the doorbell (dependent), block in host memory / in device memory.
// build: hipcc --offload-arch=gfx950 -O2 -o barprobe barprobe.hip ; run: ./barprobe
#include <hip/hip_runtime.h>
#include <csetjmp>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <immintrin.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

static sigjmp_buf jb;
static void on_fault(int) { siglongjmp(jb, 1); }

static double now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }

// a wave that answers doorbells until told to stop (seq == ~0): for every new seq it (optionally) sums `n_block` doubles of `block` and writes the
// seq (and the sum) to `done`
__global__ void pong(volatile unsigned long long *bell, const double *block, int n_block, unsigned long long *done, double *sum_out, unsigned max_polls)
{
	unsigned long long last = 0;
	for (unsigned it = 0; it < max_polls; ++it) {
		unsigned long long s = 0;
		if (threadIdx.x == 0) s = __hip_atomic_load((unsigned long long *) bell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
		s = __shfl(s, 0, 64);
		if (s == ~0ull) break;
		if (s == last) { __builtin_amdgcn_s_sleep(2); continue; }
		double acc = 0.0;
		for (int i = threadIdx.x; i < n_block; i += 64) acc += __builtin_nontemporal_load(block + i);
		for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
		last = s;
		if (threadIdx.x == 0) {
			if (n_block) __builtin_nontemporal_store(acc, sum_out);
			__hip_atomic_store(done, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

static int pingpong(const char *what, unsigned long long *bell, double *block, int n_block, bool wc)
{
	unsigned long long *done = nullptr;
	double *sum = nullptr;
	CK(hipHostMalloc((void **) &done, 64, hipHostMallocCoherent));
	CK(hipHostMalloc((void **) &sum, 64, hipHostMallocCoherent));
	*done = 0; *sum = 0;
	hipStream_t st;
	CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	*(volatile unsigned long long *) bell = 0;
	if (wc) _mm_sfence();
	hipLaunchKernelGGL(pong, dim3(1), dim3(64), 0, st, bell, block, n_block, done, sum, 1u << 24);
	CK(hipGetLastError());
	const int iters = 2000;
	double best = 1e9, total = 0.0;
	for (int i = 1; i <= iters; ++i) {
		if (block) { for (int k = 0; k < n_block; ++k) ((volatile double *) block)[k] = (double) (i + k); if (wc) _mm_sfence(); }
		const double t0 = now_us();
		__atomic_store_n(bell, (unsigned long long) i, __ATOMIC_RELEASE);
		if (wc) _mm_sfence();
		while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != (unsigned long long) i) {
			if (now_us() - t0 > 2e6) { printf("%s: no answer to doorbell %d\n", what, i); __atomic_store_n(bell, ~0ull, __ATOMIC_RELEASE); hipStreamSynchronize(st); return 1; }
		}
		const double dt = now_us() - t0;
		if (n_block) {
			double want = 0.0;
			for (int k = 0; k < n_block; ++k) want += (double) (i + k);
			if (*sum != want) { printf("%s: iteration %d: the wave summed %.1f, the block holds %.1f (stale read)\n", what, i, *sum, want); __atomic_store_n(bell, ~0ull, __ATOMIC_RELEASE); hipStreamSynchronize(st); return 1; }
		}
		if (i > 100) { total += dt; if (dt < best) best = dt; }
	}
	__atomic_store_n(bell, ~0ull, __ATOMIC_RELEASE);
	if (wc) _mm_sfence();
	CK(hipStreamSynchronize(st));
	printf("%-64s mean %6.2f us   best %6.2f us\n", what, total / (iters - 100), best);
	CK(hipStreamDestroy(st));
	CK(hipHostFree(done));
	CK(hipHostFree(sum));
	return 0;
}

int main()
{
	signal(SIGSEGV, on_fault);
	signal(SIGBUS, on_fault);
	void *fine = nullptr, *plain = nullptr, *unc = nullptr;
	hipError_t e1 = hipExtMallocWithFlags(&fine, 1 << 16, hipDeviceMallocFinegrained);
	hipError_t e2 = hipMalloc(&plain, 1 << 16);
	hipError_t e3 = hipExtMallocWithFlags(&unc, 1 << 16, hipDeviceMallocUncached);
	printf("fine-grained device memory: %s (%p); hipMalloc: %s (%p); uncached: %s (%p)\n", hipGetErrorString(e1), fine, hipGetErrorString(e2), plain, hipGetErrorString(e3), unc);
	(void) hipGetLastError();
	struct { const char *name; void *p; bool ok; } mem[3] = { { "fine-grained", fine, false }, { "hipMalloc", plain, false }, { "uncached", unc, false } };
	for (auto &m : mem) {
		if (!m.p) continue;
		if (sigsetjmp(jb, 1) == 0) {
			volatile unsigned long long *q = (volatile unsigned long long *) m.p;
			q[0] = 0x1122334455667788ull;
			q[1] = 42;
			_mm_sfence();
			unsigned long long back = 0;
			CK(hipMemcpy(&back, m.p, 8, hipMemcpyDeviceToHost));
			m.ok = back == 0x1122334455667788ull;
			printf("CPU store into %s device memory: no fault; the device holds %s\n", m.name, m.ok ? "what was stored" : "something else");
			if (m.ok) { const unsigned long long r = q[1]; printf("CPU load from it: %llu (%s)\n", r, r == 42 ? "ok" : "wrong"); }
		}
		else printf("CPU store into %s device memory: fault (not host-accessible)\n", m.name);
	}
	// ping-pong, doorbell and block in host memory (today)
	unsigned long long *hbell = nullptr;
	double *hblock = nullptr;
	CK(hipHostMalloc((void **) &hbell, 64, hipHostMallocCoherent));
	CK(hipHostMalloc((void **) &hblock, 4096, hipHostMallocCoherent));
	if (pingpong("doorbell in host memory, no block", hbell, nullptr, 0, false)) return 1;
	if (pingpong("doorbell in host memory, 1 KB block in host memory", hbell, hblock, 128, false)) return 1;
	if (pingpong("doorbell in host memory, 4 KB block in host memory", hbell, hblock, 512, false)) return 1;
	for (auto &m : mem) {
		if (!m.ok) continue;
		char what[128];
		unsigned long long *dbell = (unsigned long long *) m.p;
		double *dblock = (double *) ((char *) m.p + 4096);
		snprintf(what, sizeof what, "doorbell in %s device memory, no block", m.name);
		if (pingpong(what, dbell, nullptr, 0, true)) continue;
		snprintf(what, sizeof what, "doorbell + 1 KB block in %s device memory", m.name);
		if (pingpong(what, dbell, dblock, 128, true)) continue;
		snprintf(what, sizeof what, "doorbell + 4 KB block in %s device memory", m.name);
		if (pingpong(what, dbell, dblock, 512, true)) continue;
	}
	return 0;
}
