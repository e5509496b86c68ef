// kernels_chunk.hip -- the time axis of FEW channels spread over the whole chip.
//
// A biquad cascade is linear in (input, state): with D = 2 n_ops state variables per channel,
//     state after a chunk of len frames   x' = M x + e        (M: D x D, e: what the chunk leaves behind when started from zero)
//     output of the chunk                 y  = y0 + H x        (y0: zero-state output, H: len x D, row i = response at frame i)
// M and H depend only on the sections and on len (host, extended precision: CascadeStage::build_chunk_plan).  So a call of
// K len frames on S streams is run as S K independent "streams" of len frames from ZERO state by the ordinary cascade
// kernels (the chip sees S K C channels instead of S C: one 8-channel stream becomes 3072 channels), then
//   cascade_chunk_carry   per (stream, channel): e_0 += M x_0, two-level scan over the K chunks with M, M^2 .. M^g -> the true
//                         state x_c at the start of every chunk, and the state carried to the next call
//   cascade_chunk_fix     y[c][i] += H[i] . x_c   (D FMAs per sample, one more pass over the output)
// The reference's recurrence (biquad.h:76-92) is the K = 1 case; the results agree to rounding (1e-15 relative).
#include <hip/hip_runtime.h>
#include "kparams.h"

namespace dspamd {

// grid (S, C), block 1024; LDS: K D + G D + D D doubles.  Mp: [n_cls][g][D][D] (M^(j+1), j = 0 .. g-1, row-major), cls: [C].
// Two-level scan over the K chunks in groups of g (about sqrt(K)): g sequential steps inside all groups at once, G = ceil(K/g)
// sequential steps over the group ends, one parallel step that hands every chunk its group's carry -- 2 K + G matrix-vector
// products instead of K log K, and no step touches more than one power of M.
template <int DH>   // D / 2 at compile time (the dot products unroll: their loads overlap), 0 = any D
__global__ __launch_bounds__(1024) void cascade_chunk_carry(ChunkParams p)
{
	extern __shared__ __attribute__((aligned(16))) double smem[];
	const int D = DH ? 2 * DH : p.D;
	const int s = blockIdx.x, ch = blockIdx.y, K = p.K, g = p.n_pow, G = (K + g - 1) / g, n = K * D;
	const double *Mp = p.Mp + (size_t) p.cls[ch] * g * D * D;
	double *a = smem, *carry = smem + n, *mt = carry + G * D;  // mt: a matrix TRANSPOSED (consecutive rows = consecutive banks)
	double *x0 = p.state + ((size_t) s * p.C + ch) * D;
	const int tid = threadIdx.x;
	for (int e = tid; e < n; e += blockDim.x) {
		const int c = e / D, r = e - c * D;
		a[e] = p.cstate[(((size_t) s * K + c) * p.C + ch) * D + r];
	}
	for (int e = tid; e < D * D; e += blockDim.x) { const int r = e / D, k = e - r * D; mt[k * D + r] = Mp[e]; }      // M
	__syncthreads();
	// the state carried into the call enters through chunk 0: e_0 += M x_0
	double v = 0.0;
	if (tid < D) {
		v = a[tid];
#pragma unroll
		for (int k = 0; k < D; ++k) v = fma(mt[k * D + tid], x0[k], v);
	}
	__syncthreads();
	if (tid < D) a[tid] = v;
	__syncthreads();
	// level 1: inside every group, T[c] = M T[c - 1] + e[c]
	// Where a group's D rows fit half a wave and the groups fit the workgroup (D <= 32, G <= 32: every plan the host makes for up to twelve sections and a
	// thousand chunks), group q is lanes 32 (q mod 2) .. of wave q / 2: a step's results are read back by the SAME wave, whose LDS traffic is processed in
	// order -- the g steps need no workgroup barrier (two per step until round 6: 32 of config 2's 109 us per call were this kernel's barriers).
	const bool wave_groups = (D <= 32) && (G <= 32) && (blockDim.x == 1024);
	auto wave_sync = [] {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	};
	if (wave_groups) {
		const int q1 = tid >> 5, r1 = tid & 31;
		const bool on = q1 < G && r1 < D;
		for (int j = 1; j < g; ++j) {
			const int c = q1 * g + j;
			if (on && c < K) {
				const double *src = a + (c - 1) * D;
				double acc = a[c * D + r1], acc2 = 0.0;
#pragma unroll
				for (int k = 0; k + 1 < D; k += 2) { acc = fma(mt[k * D + r1], src[k], acc); acc2 = fma(mt[(k + 1) * D + r1], src[k + 1], acc2); }
				a[c * D + r1] = acc + acc2;
			}
			wave_sync();
		}
		__syncthreads();
	}
	else {
		const int q1 = tid / D, r1 = tid - q1 * D;                  // thread = (group, row)
		for (int j = 1; j < g; ++j) {
			const int c = q1 * g + j;
			if (q1 < G && c < K) {
				const double *src = a + (c - 1) * D;
				double acc = a[c * D + r1], acc2 = 0.0;
#pragma unroll
				for (int k = 0; k + 1 < D; k += 2) { acc = fma(mt[k * D + r1], src[k], acc); acc2 = fma(mt[(k + 1) * D + r1], src[k + 1], acc2); }
				v = acc + acc2;
			}
			__syncthreads();
			if (q1 < G && c < K) a[c * D + r1] = v;
			__syncthreads();
		}
	}
	// level 2: carry into group q = true end state of group q - 1 = T[end of q - 1] + M^g carry[q - 1]
	for (int e = tid; e < D * D; e += blockDim.x) { const int r = e / D, k = e - r * D; mt[k * D + r] = Mp[(size_t) (g - 1) * D * D + e]; }   // M^g
	if (tid < D) carry[tid] = 0.0;
	__syncthreads();
	if (D <= 64) {
		// (the D rows are lanes of wave 0: its steps order themselves)
		if (tid < 64) {
			for (int q = 1; q < G; ++q) {
				if (tid < D) {
					const double *src = carry + (q - 1) * D;
					double acc = a[(q * g - 1) * D + tid];
#pragma unroll
					for (int k = 0; k < D; ++k) acc = fma(mt[k * D + tid], src[k], acc);
					carry[q * D + tid] = acc;
				}
				wave_sync();
			}
		}
		__syncthreads();
	}
	else {
		for (int q = 1; q < G; ++q) {
			if (tid < D) {
				const double *src = carry + (q - 1) * D;
				double acc = a[(q * g - 1) * D + tid];
#pragma unroll
				for (int k = 0; k < D; ++k) acc = fma(mt[k * D + tid], src[k], acc);
				carry[q * D + tid] = acc;
			}
			__syncthreads();
		}
	}
	// level 3: T[c] += M^(j + 1) carry[q], c = q g + j; then T[c] = state at the END of chunk c = at the start of chunk c + 1
	for (int e = tid; e < n; e += blockDim.x) {
		const int c = e / D, r = e - c * D, q = c / g, j = c - q * g;
		double acc = a[e];
		if (q > 0) {
			const double *m = Mp + ((size_t) j * D + r) * D, *src = carry + q * D;
#pragma unroll
			for (int k = 0; k < D; ++k) acc = fma(m[k], src[k], acc);
		}
		if (c + 1 < K) p.X[(((size_t) s * K + c + 1) * p.C + ch) * D + r] = acc;
		else v = acc;                                           // (the thread that holds the last chunk's row r)
		if (c == 0) p.X[(((size_t) s * K) * p.C + ch) * D + r] = x0[r];
	}
	__syncthreads();              // every x0 has been read
	for (int e = tid; e < n; e += blockDim.x) if (e / D == K - 1) x0[e - (K - 1) * D] = v;
}

// All channels share one table (the usual case): a lane owns a COLUMN (chunk, channel) -- its state in registers -- and walks
// down a strip of frames, so that the response row H[i] is the same for the whole wave and comes through scalar loads:
// per sample one load, D FMAs with an SGPR operand, one store.  grid (ceil(S K C / 256), len / FIX_STRIP), block 256.
constexpr int FIX_STRIP = 64;
template <int DH>    // DH = D / 2 when it is small enough to keep the state in registers
__global__ __launch_bounds__(256) void cascade_chunk_fix_cols(ChunkParams p, const double *__restrict__ H, int n_cols)   // H = p.H (noalias: scalar loads)
{
	const int col = blockIdx.x * blockDim.x + threadIdx.x;        // (s K + c) C + ch
	if (col >= n_cols) return;
	const int v = col / p.C, ch = col - v * p.C;
	const int s = v / p.K, c = v - s * p.K;
	const long i0 = (long) blockIdx.y * FIX_STRIP;
	double2 x[DH];
	const double2 *xs = reinterpret_cast<const double2 *>(p.X + (size_t) col * (2 * DH));
#pragma unroll
	for (int k = 0; k < DH; ++k) x[k] = xs[k];
	double *y = p.out + ((size_t) s * p.out_stride_frames + (size_t) c * p.len + i0) * p.C + ch;
	const double *__restrict__ h = H + (size_t) i0 * (2 * DH);     // wave-uniform
#pragma unroll 4
	for (int i = 0; i < FIX_STRIP; ++i) {
		double a0 = 0.0, a1 = 0.0;
#pragma unroll
		for (int k = 0; k < DH; ++k) { a0 = fma(h[(size_t) i * (2 * DH) + 2 * k], x[k].x, a0); a1 = fma(h[(size_t) i * (2 * DH) + 2 * k + 1], x[k].y, a1); }
		y[(size_t) i * p.C] += a0 + a1;
	}
}

// general form (channels with different sections): one thread per (frame, channel) of a chunk, everything through the caches.
// grid (ceil(len C / 256), S K), block 256.  H: [n_cls][len][D]
__global__ __launch_bounds__(256) void cascade_chunk_fix(ChunkParams p)
{
	const int v = blockIdx.y, D = p.D;
	const long e = (long) blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= p.len * p.C) return;
	const long i = e / p.C;
	const int ch = (int) (e - i * p.C);
	const int s = v / p.K, c = v - s * p.K;
	const double2 *h = reinterpret_cast<const double2 *>(p.H + ((size_t) p.cls[ch] * p.len + i) * D);
	const double2 *x = reinterpret_cast<const double2 *>(p.X + ((size_t) v * p.C + ch) * D);
	double a0 = 0.0, a1 = 0.0;
	for (int k = 0; k < D / 2; ++k) {
		const double2 hv = h[k], xv = x[k];
		a0 = fma(hv.x, xv.x, a0);
		a1 = fma(hv.y, xv.y, a1);
	}
	double *y = p.out + ((size_t) s * p.out_stride_frames + (size_t) c * p.len + i) * p.C + ch;
	*y += a0 + a1;
}

template <int DH> static void launch_carry(const ChunkParams &p, int n_streams, size_t lds, hipStream_t stream)
{
	grant_dynamic_lds(reinterpret_cast<const void *>(cascade_chunk_carry<DH>), lds);
	hipLaunchKernelGGL(cascade_chunk_carry<DH>, dim3(n_streams, p.C), dim3(1024), lds, stream, p);
}

void launch_chunk_carry(const ChunkParams &p, int n_streams, hipStream_t stream)
{
	const int G = (p.K + p.n_pow - 1) / p.n_pow;
	const size_t lds = ((size_t) p.K * p.D + (size_t) G * p.D + (size_t) p.D * p.D) * sizeof(double);
	switch (p.D / 2) {
	case 1: return launch_carry<1>(p, n_streams, lds, stream);
	case 2: return launch_carry<2>(p, n_streams, lds, stream);
	case 3: return launch_carry<3>(p, n_streams, lds, stream);
	case 4: return launch_carry<4>(p, n_streams, lds, stream);
	case 5: return launch_carry<5>(p, n_streams, lds, stream);
	case 6: return launch_carry<6>(p, n_streams, lds, stream);
	case 7: return launch_carry<7>(p, n_streams, lds, stream);
	case 8: return launch_carry<8>(p, n_streams, lds, stream);
	case 9: return launch_carry<9>(p, n_streams, lds, stream);
	case 10: return launch_carry<10>(p, n_streams, lds, stream);
	case 11: return launch_carry<11>(p, n_streams, lds, stream);
	case 12: return launch_carry<12>(p, n_streams, lds, stream);
	default: return launch_carry<0>(p, n_streams, lds, stream);
	}
}

template <int DH> static void launch_fix_cols(const ChunkParams &p, int n_streams, hipStream_t stream)
{
	const int n_cols = n_streams * p.K * p.C;
	hipLaunchKernelGGL(cascade_chunk_fix_cols<DH>, dim3((unsigned) ((n_cols + 255) / 256), (unsigned) (p.len / FIX_STRIP)), dim3(256), 0, stream, p, p.H, n_cols);
}

void launch_chunk_fix(const ChunkParams &p, int n_streams, hipStream_t stream)
{
	if (p.n_cls == 1 && p.len % FIX_STRIP == 0) {
		switch (p.D / 2) {
		case 1: return launch_fix_cols<1>(p, n_streams, stream);
		case 2: return launch_fix_cols<2>(p, n_streams, stream);
		case 3: return launch_fix_cols<3>(p, n_streams, stream);
		case 4: return launch_fix_cols<4>(p, n_streams, stream);
		case 5: return launch_fix_cols<5>(p, n_streams, stream);
		case 6: return launch_fix_cols<6>(p, n_streams, stream);
		case 7: return launch_fix_cols<7>(p, n_streams, stream);
		case 8: return launch_fix_cols<8>(p, n_streams, stream);
		case 9: return launch_fix_cols<9>(p, n_streams, stream);
		case 10: return launch_fix_cols<10>(p, n_streams, stream);
		case 11: return launch_fix_cols<11>(p, n_streams, stream);
		case 12: return launch_fix_cols<12>(p, n_streams, stream);
		case 13: return launch_fix_cols<13>(p, n_streams, stream);
		case 14: return launch_fix_cols<14>(p, n_streams, stream);
		case 15: return launch_fix_cols<15>(p, n_streams, stream);
		case 16: return launch_fix_cols<16>(p, n_streams, stream);
		default: break;
		}
	}
	const long blocks = (p.len * p.C + 255) / 256;
	hipLaunchKernelGGL(cascade_chunk_fix, dim3((unsigned) blocks, (unsigned) (n_streams * p.K)), dim3(256), 0, stream, p);
}

}  // namespace dspamd
