// hbmprobe.hip -- what the memory system of this MI355X gives to the access patterns the convolver uses.
//   1. plain streaming: copy / read-only / write-only, U 16-byte accesses in flight per lane, non-temporal variants,
//      grid sizes from one workgroup per CU up to one tile per workgroup  -> the box's HBM ceiling
//   2. the column pattern of K1 / K3 (conv_col_fwd / conv_col_inv): 16 rows x (TW x 16 B) runs at a row pitch of
//      N2 x 16 B, with and without a padded pitch, with 256- and 512-thread workgroups
//   3. the row pattern of K2 (conv_row): a wave owns a contiguous 1024-point row, in place
// No arithmetic: what these loops reach is the ceiling of the pattern itself.
// Build: hipcc -O3 --offload-arch=gfx950 hbmprobe.hip -o hbmprobe        Run: ./hbmprobe [GiB per buffer = 4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>

typedef double2 e16;

template <bool NT> __device__ __forceinline__ e16 ld(const e16 *p)
{
	if (NT) { e16 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }
	return *p;
}
template <bool NT> __device__ __forceinline__ void st(e16 *p, e16 v)
{
	if (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
	else *p = v;
}

// MODE 0 copy, 1 read, 2 write.  A workgroup walks tiles of 256 * U elements: tile b, b + grid, ...
template <int U, int MODE, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_stream(const e16 *__restrict__ src, e16 *__restrict__ dst, size_t n_tiles, double *sink)
{
	e16 acc = make_double2(0, 0);
	for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const size_t base = tile * (256 * U) + threadIdx.x;
		e16 v[U];
		if (MODE != 2) {
#pragma unroll
			for (int u = 0; u < U; ++u) v[u] = ld<NTL>(src + base + 256 * u);
		}
		else {
#pragma unroll
			for (int u = 0; u < U; ++u) v[u] = make_double2((double) tile, (double) u);
		}
		if (MODE != 1) {
#pragma unroll
			for (int u = 0; u < U; ++u) st<NTS>(dst + base + 256 * u, v[u]);
		}
		else {
#pragma unroll
			for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; }
		}
	}
	if (MODE == 1 && acc.x == 1.2345) *sink = acc.y;
}

// column pattern: grid (N2 / TW, pairs); thread (t, j) moves rows j + P m, m < 16, column blockIdx.x * TW + t
template <int THREADS, int N1>
__global__ __launch_bounds__(THREADS) void k_col(const e16 *__restrict__ src, e16 *__restrict__ dst, long src_pitch, long dst_pitch,
                                                 long src_pair, long dst_pair, int do_load, int do_store)
{
	constexpr int P = N1 / 16, TW = THREADS / P;
	const int t = threadIdx.x % TW, j = threadIdx.x / TW;
	const long n2 = (long) blockIdx.x * TW + t;
	const e16 *s = src + blockIdx.y * src_pair + n2;
	e16 *d = dst + blockIdx.y * dst_pair + n2;
	e16 v[16];
#pragma unroll
	for (int m = 0; m < 16; ++m) v[m] = do_load ? s[(long) (j + P * m) * src_pitch] : make_double2((double) j, (double) m);
	if (do_store) {
#pragma unroll
		for (int m = 0; m < 16; ++m) d[(long) (j + P * m) * dst_pitch] = v[m];
	}
	else {
		e16 a = make_double2(0, 0);
#pragma unroll
		for (int m = 0; m < 16; ++m) { a.x += v[m].x; a.y += v[m].y; }
		if (a.x == 1.2345) d[0] = a;
	}
}

// K3 as it is launched: 512 threads = 4 pairs (lane-fastest) x 8 columns x 16 row slots; reads 16 rows j + 16 m of every pair
// (128 B runs per pair), writes 8 consecutive 64-byte frames per row (512 B runs) into an interleaved slab
__global__ __launch_bounds__(512) void k_k3(const e16 *__restrict__ W, e16 *__restrict__ out, long n2_total, long pitch, long pair_stride, int do_store)
{
	const int q = threadIdx.x & 3, t = (threadIdx.x >> 2) & 7, j = threadIdx.x >> 5;
	const long n2 = (long) blockIdx.x * 8 + t;
	const e16 *s = W + ((long) blockIdx.y * 4 + q) * pair_stride + n2;
	e16 v[16];
#pragma unroll
	for (int m = 0; m < 16; ++m) v[m] = s[(long) (j + 16 * m) * pitch];
	if (do_store) {
		e16 *d = out + (long) blockIdx.y * 4 * 256 * n2_total + q;       // [stream][frame][4 pairs]
#pragma unroll
		for (int m = 0; m < 16; ++m) d[((long) (j + 16 * m) * n2_total + n2) * 4] = v[m];
	}
	else {
		e16 a = make_double2(0, 0);
#pragma unroll
		for (int m = 0; m < 16; ++m) { a.x += v[m].x; a.y += v[m].y; }
		if (a.x == 1.2345) out[0] = a;
	}
}

// K1 reading an INTERLEAVED slab (frames of 64 bytes = 4 pairs) directly: PPS pairs per workgroup, lanes pair-fastest, so a
// frame contributes PPS * 16 contiguous bytes; 16 columns x 256 rows per pair, W-like destination (256-byte runs per pair)
template <int PPS>
__global__ __launch_bounds__(256 * PPS) void k_slab(const e16 *__restrict__ slab, e16 *__restrict__ W, long n2_total)
{
	const int q = threadIdx.x % PPS, t = (threadIdx.x / PPS) % 16, j = threadIdx.x / (16 * PPS);
	const long n2 = (long) blockIdx.x * 16 + t;
	const long stream = blockIdx.y / (4 / PPS), pq = (blockIdx.y % (4 / PPS)) * PPS + q;
	const e16 *s = slab + stream * 256 * n2_total * 4 + pq;
	e16 *d = W + (stream * 4 + pq) * (256 * n2_total + 272) + n2;
	e16 v[16];
#pragma unroll
	for (int m = 0; m < 16; ++m) v[m] = s[((long) (j + 16 * m) * n2_total + n2) * 4];
#pragma unroll
	for (int m = 0; m < 16; ++m) d[(long) (j + 16 * m) * n2_total] = v[m];
}

// row pattern: grid (N1 / 4, pairs): 4 rows of 1024 points per workgroup, one per wave; lane moves j + 64 m
__global__ __launch_bounds__(256) void k_row(const e16 *__restrict__ src, e16 *__restrict__ dst, long pitch, long pair_stride)
{
	const int rw = threadIdx.x >> 6, j = threadIdx.x & 63;
	const long k1 = (long) blockIdx.x * 4 + rw;
	const e16 *s = src + blockIdx.y * pair_stride + k1 * pitch + j;
	e16 *d = dst + blockIdx.y * pair_stride + k1 * pitch + j;
	e16 v[16];
#pragma unroll
	for (int m = 0; m < 16; ++m) v[m] = s[64 * m];
#pragma unroll
	for (int m = 0; m < 16; ++m) d[64 * m] = v[m];
}

static hipEvent_t e0, e1;
template <class F> static double time_ms(F f, int reps)
{
	f();
	hipEventRecord(e0);
	for (int r = 0; r < reps; ++r) f();
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	return ms / reps;
}

int main(int argc, char **argv)
{
	const size_t gib = argc > 1 ? (size_t) atoi(argv[1]) : 4;
	const size_t bytes = gib << 30, slack = (size_t) 256 << 20;
	e16 *a, *b;
	double *sink;
	if (hipMalloc(&a, bytes + slack) != hipSuccess || hipMalloc(&b, bytes + slack) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipMalloc(&sink, 8);
	hipMemset(a, 0, bytes + slack);
	hipMemset(b, 0, bytes + slack);
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int reps = 6;
	printf("# plain streaming, %zu GiB per buffer (TB/s of bytes moved: copy counts read + write)\n", gib);
	printf("%-34s %8s %8s %8s\n", "variant", "copy", "read", "write");
#define STREAM_ROW(U, NTL, NTS, GRID, LABEL)                                                                                   \
	{                                                                                                                          \
		const size_t nt = bytes / 16 / (256 * U);                                                                              \
		const unsigned g = (GRID) ? (unsigned) (GRID) : (unsigned) nt;                                                         \
		const double c = time_ms([&] { k_stream<U, 0, NTL, NTS><<<g, 256>>>(a, b, nt, sink); }, reps);                          \
		const double r = time_ms([&] { k_stream<U, 1, NTL, NTS><<<g, 256>>>(a, b, nt, sink); }, reps);                          \
		const double w = time_ms([&] { k_stream<U, 2, NTL, NTS><<<g, 256>>>(a, b, nt, sink); }, reps);                          \
		printf("%-34s %8.2f %8.2f %8.2f\n", LABEL, 2.0 * bytes / c / 1e9, 1.0 * bytes / r / 1e9, 1.0 * bytes / w / 1e9);        \
	}
	STREAM_ROW(1, false, false, 2048, "U=1 grid 2048 (round-1 probe)")
	STREAM_ROW(1, false, false, 0, "U=1 one tile per wg")
	STREAM_ROW(2, false, false, 0, "U=2 one tile per wg")
	STREAM_ROW(4, false, false, 0, "U=4 one tile per wg")
	STREAM_ROW(8, false, false, 0, "U=8 one tile per wg")
	STREAM_ROW(16, false, false, 0, "U=16 one tile per wg")
	STREAM_ROW(4, false, false, 1024, "U=4 grid 1024")
	STREAM_ROW(4, false, false, 2048, "U=4 grid 2048")
	STREAM_ROW(4, false, false, 4096, "U=4 grid 4096")
	STREAM_ROW(8, false, false, 2048, "U=8 grid 2048")
	STREAM_ROW(8, false, false, 4096, "U=8 grid 4096")
	STREAM_ROW(4, true, false, 0, "U=4 nt loads")
	STREAM_ROW(4, false, true, 0, "U=4 nt stores")
	STREAM_ROW(4, true, true, 0, "U=4 nt loads+stores")
	STREAM_ROW(8, true, true, 0, "U=8 nt loads+stores")
	STREAM_ROW(8, true, true, 4096, "U=8 nt both grid 4096")

	// column pattern on 1024 "pairs" x 256 rows x 1024 points (the headline geometry, 4 GiB per buffer)
	printf("# column pattern (K1 / K3): 256 rows x 1024 points per pair, pairs = %zu; TB/s of bytes moved\n", bytes / ((size_t) 256 * 1024 * 16));
	printf("%-44s %8s %8s %8s\n", "variant", "copy", "load", "store");
	const int pairs = (int) (bytes / ((size_t) 256 * 1024 * 16));
	const double colbytes = (double) pairs * 256 * 1024 * 16;
	struct { long pitch_pad, pair_pad; const char *label; } pads[] = {
		{ 0, 0, "pitch 1024 (as round 1)" },        { 16, 0, "pitch 1024+16 (256 B)" },   { 32, 0, "pitch 1024+32 (512 B)" },
		{ 8, 0, "pitch 1024+8 (128 B)" },           { 64, 0, "pitch 1024+64 (1 KiB)" },   { 0, 16 * 17, "pitch 1024, pair +4352 B" },
		{ 16, 16 * 17, "pitch +256 B, pair +4352 B" },
	};
	for (auto &pd : pads) {
		const long pitch = 1024 + pd.pitch_pad, pair = 256 * pitch + pd.pair_pad;
		if ((size_t) pair * pairs * 16 > bytes + slack) continue;
		for (int wg = 0; wg < 2; ++wg) {
			double t[3];
			for (int mode = 0; mode < 3; ++mode) {
				const int dl = mode != 2, ds = mode != 1;
				if (wg == 0) t[mode] = time_ms([&] { k_col<256, 256><<<dim3(1024 / 16, pairs), 256>>>(a, b, pitch, pitch, pair, pair, dl, ds); }, reps);
				else t[mode] = time_ms([&] { k_col<512, 256><<<dim3(1024 / 32, pairs), 512>>>(a, b, pitch, pitch, pair, pair, dl, ds); }, reps);
			}
			char lab[96];
			snprintf(lab, sizeof lab, "%s, %s", pd.label, wg ? "512 thr (512 B runs)" : "256 thr (256 B runs)");
			printf("%-44s %8.2f %8.2f %8.2f\n", lab, 2 * colbytes / t[0] / 1e9, colbytes / t[1] / 1e9, colbytes / t[2] / 1e9);
		}
	}
	// contiguous source (the ring as K1 reads it: stride N2 between rows too, the destination padded)
	{
		const long pitch = 1024 + 16;
		const double t = time_ms([&] { k_col<256, 256><<<dim3(1024 / 16, pairs), 256>>>(a, b, 1024, pitch, 256 * 1024, 256 * pitch, 1, 1); }, reps);
		printf("%-44s %8.2f\n", "src pitch 1024 -> dst pitch 1024+16, 256 thr", 2 * colbytes / t / 1e9);
	}
	printf("# row pattern (K2), in place and out of place; TB/s of bytes moved\n");
	for (long pad : { 0L, 16L, 32L }) {
		const long pitch = 1024 + pad, pair = 256 * pitch;
		const double t0 = time_ms([&] { k_row<<<dim3(64, pairs), 256>>>(a, a, pitch, pair); }, reps);
		const double t1 = time_ms([&] { k_row<<<dim3(64, pairs), 256>>>(a, b, pitch, pair); }, reps);
		printf("pitch 1024+%-3ld  in place %8.2f   a->b %8.2f\n", pad, 2 * colbytes / t0 / 1e9, 2 * colbytes / t1 / 1e9);
	}
	// K3's read pattern: 128-byte runs (8 columns) of 256 rows at the row pitch of the transform, 4 pairs per 512-thread workgroup
	// (pairs 4 MiB x pitch / 1024 apart); load only
	printf("# K3 read pattern: 128 B runs x 256 rows, pitch = N2 (+pad) points; TB/s\n");
	for (long n2 : { 1024L, 2048L, 4096L })
		for (long pad : { 0L, 16L, 32L }) {
			const long pitch = n2 + pad, pair = 256 * pitch;
			const int np = (int) (bytes / ((size_t) pair * 16));
			const double t = time_ms([&] { k_col<128, 256><<<dim3((unsigned) (n2 / 8), np), 128>>>(a, b, pitch, pitch, pair, pair, 1, 0); }, reps);
			const double t2 = time_ms([&] { k_col<256, 256><<<dim3((unsigned) (n2 / 16), np), 256>>>(a, b, pitch, pitch, pair, pair, 1, 0); }, reps);
			printf("N2 %5ld pad %3ld : 128 B runs %6.2f   256 B runs %6.2f\n", n2, pad, (double) np * 256 * n2 * 16 / t / 1e9, (double) np * 256 * n2 * 16 / t2 / 1e9);
		}
	// K1's WRITE pattern (round 4: would 4-column tiles do?): 64- / 128- / 256-byte runs x 256 rows at the row pitch of the transform; store only
	printf("# K1 write pattern: runs x 256 rows, pitch = N2 + 16 points; store-only TB/s\n");
	for (long n2 : { 1024L, 4096L }) {
		const long pitch = n2 + 16, pair = 256 * pitch + 272;
		const int np = (int) (bytes / ((size_t) pair * 16));
		const double t4 = time_ms([&] { k_col<64, 256><<<dim3((unsigned) (n2 / 4), np), 64>>>(a, b, pitch, pitch, pair, pair, 0, 1); }, reps);
		const double t8 = time_ms([&] { k_col<128, 256><<<dim3((unsigned) (n2 / 8), np), 128>>>(a, b, pitch, pitch, pair, pair, 0, 1); }, reps);
		const double t16 = time_ms([&] { k_col<256, 256><<<dim3((unsigned) (n2 / 16), np), 256>>>(a, b, pitch, pitch, pair, pair, 0, 1); }, reps);
		const double by = (double) np * 256 * n2 * 16;
		printf("N2 %5ld : 64 B runs %6.2f   128 B runs %6.2f   256 B runs %6.2f\n", n2, by / t4 / 1e9, by / t8 / 1e9, by / t16 / 1e9);
	}
	printf("# K3 as launched (512 threads, 4 pairs, 128 B read runs, 512 B write runs): TB/s of bytes moved\n");
	for (long n2 : { 1024L, 2048L, 4096L })
		for (long pad : { 0L, 16L })
			for (long ppad : { 0L, 272L }) {
				const long pitch = n2 + pad, pair = 256 * pitch + ppad;
				const int ns = (int) (bytes / ((size_t) pair * 16 * 4));
				const double tl = time_ms([&] { k_k3<<<dim3((unsigned) (n2 / 8), ns), 512>>>(a, b, n2, pitch, pair, 0); }, reps);
				const double tc = time_ms([&] { k_k3<<<dim3((unsigned) (n2 / 8), ns), 512>>>(a, b, n2, pitch, pair, 1); }, reps);
				const double by = (double) ns * 4 * 256 * n2 * 16;
				printf("N2 %5ld pitch pad %3ld pair pad %4ld : load %6.2f   copy %6.2f\n", n2, pad, ppad, by / tl / 1e9, 2 * by / tc / 1e9);
			}
	printf("# K1 reading an interleaved slab directly (4 pairs per 64-byte frame): pairs per workgroup; TB/s of bytes moved\n");
	for (long n2 : { 1024L, 4096L }) {
		const int ns = (int) (bytes / ((size_t) 256 * n2 * 64));
		const double by = 2.0 * ns * 256 * n2 * 64;
		const double t1 = time_ms([&] { k_slab<1><<<dim3((unsigned) (n2 / 16), ns * 4), 256>>>(a, b, n2); }, reps);
		const double t2 = time_ms([&] { k_slab<2><<<dim3((unsigned) (n2 / 16), ns * 2), 512>>>(a, b, n2); }, reps);
		const double t4 = time_ms([&] { k_slab<4><<<dim3((unsigned) (n2 / 16), ns), 1024>>>(a, b, n2); }, reps);
		printf("N2 %5ld : 1 pair / wg (16 B pieces) %6.2f   2 pairs (32 B) %6.2f   4 pairs (whole frames, 1024 threads) %6.2f\n", n2, by / t1 / 1e9, by / t2 / 1e9, by / t4 / 1e9);
	}
	// does the rate depend on how many workgroups (= bytes in flight) a CU holds?  dynamic LDS caps the residency
	printf("# residency sweep: workgroups per CU capped through dynamic LDS (copy TB/s)\n");
	printf("%-10s %12s %12s %12s %12s\n", "LDS/wg", "stream U=16", "stream U=4", "col 256thr", "row in place");
	for (size_t lds : { (size_t) 0, (size_t) 20 << 10, (size_t) 32 << 10, (size_t) 40 << 10, (size_t) 52 << 10, (size_t) 64 << 10, (size_t) 80 << 10, (size_t) 160 << 10 }) {
		hipFuncSetAttribute((const void *) k_stream<16, 0, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
		hipFuncSetAttribute((const void *) k_stream<4, 0, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
		hipFuncSetAttribute((const void *) k_col<256, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
		hipFuncSetAttribute((const void *) k_row, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
		const size_t nt16 = bytes / 16 / (256 * 16), nt4 = bytes / 16 / (256 * 4);
		const double a16 = time_ms([&] { hipLaunchKernelGGL((k_stream<16, 0, false, false>), dim3((unsigned) nt16), dim3(256), lds, 0, a, b, nt16, sink); }, reps);
		const double a4 = time_ms([&] { hipLaunchKernelGGL((k_stream<4, 0, false, false>), dim3((unsigned) nt4), dim3(256), lds, 0, a, b, nt4, sink); }, reps);
		const double c = time_ms([&] { hipLaunchKernelGGL((k_col<256, 256>), dim3(1024 / 16, pairs), dim3(256), lds, 0, a, b, 1024L, 1024L, 256L * 1024, 256L * 1024, 1, 1); }, reps);
		const double r = time_ms([&] { hipLaunchKernelGGL(k_row, dim3(64, pairs), dim3(256), lds, 0, a, a, 1024L, 256L * 1024); }, reps);
		printf("%-10zu %12.2f %12.2f %12.2f %12.2f\n", lds, 2.0 * bytes / a16 / 1e9, 2.0 * bytes / a4 / 1e9, 2 * colbytes / c / 1e9, 2 * colbytes / r / 1e9);
	}
	return 0;
}
