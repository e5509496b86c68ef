// kparams.h -- plain-data launch parameters shared by the HIP kernels and the host engine.
#pragma once
#include <cstddef>
#include <cstdint>
#include <sys/types.h>

namespace dspamd {

// ---- wire formats at either end of a pipeline ----
// same numbering as DSPAMD_PCM_* in include/dsp_amd.h
enum { PCM_U8 = 0, PCM_S8, PCM_S16, PCM_S24, PCM_S32, PCM_S24_3, PCM_FLOAT, PCM_DOUBLE, PCM_N_FORMATS };
inline size_t pcm_sample_bytes(int fmt)
{
	constexpr size_t b[PCM_N_FORMATS] = { 1, 1, 2, 4, 4, 3, 4, 8 };
	return (fmt >= 0 && fmt < PCM_N_FORMATS) ? b[fmt] : 0;
}
// formats a kernel can take / leave in its own loads and stores: naturally aligned channel pairs (the others go through
// the stand-alone conversion kernels)
inline bool pcm_fusable(int fmt) { return fmt == PCM_S16 || fmt == PCM_S24 || fmt == PCM_S32 || fmt == PCM_FLOAT || fmt == PCM_DOUBLE; }

// The output stage of dsp.c:685-699 done by the LAST kernel of a pipeline in its stores: [TPDF dither], clip() with its
// statistics, write_buf_<fmt>.  `on` = 0: the kernel writes plain fp64 samples as usual.
struct WireSink {
	int on, fmt;
	double dither_mult;        // 0 = no dither, else 1 / (PM_RAND_MAX 2^(prec-1))  (util.h:157-163)
	long samples_before;       // samples of each stream written before frame 0 of the destination (position in the dither sequence)
	double *stats;             // optional [S][2]: clipped samples (as a 64-bit count), peak |sample|
};

// num_records of a raw buffer descriptor (__builtin_amdgcn_make_buffer_rsrc): the bytes a kernel is MEANT to touch from the descriptor's base.  Beyond
// them the hardware answers a load with zeros and drops a store -- an access outside the buffer becomes a wrong VALUE a parity test sees, not a read of
// whatever the allocator put next to the buffer (or of nothing: a memory fault that depends on the allocator's mood).  Until round 6 every descriptor
// said 0x7fffffff: the check was switched off.  Descriptors address at most 2 GB (32-bit offsets: the host keeps rings, slabs and outputs below that).
constexpr int rsrc_records(long bytes) { return bytes <= 0 ? 0 : (bytes > 0x7fffffffL ? 0x7fffffff : (int) bytes); }

// ---- cascade (gain / add / biquad sections fused into one pass) ----

enum : int { OP_MUL = 0, OP_ADD = 1, OP_BIQUAD = 2, OP_SKIP = 3 };

constexpr int CASCADE_L = 16;            // samples per lane in a full tile
constexpr int CASCADE_TILE = 64 * CASCADE_L;
constexpr int CASCADE_NPOW = 11;         // A^(2^k), k = 0..10 (covers L*2^5 = 512 = 2^9 and below)

// One operation on one channel.  Wave-uniform, read through the scalar cache.
struct alignas(16) OpDesc {
	int kind, pad;
	double g;                            // OP_MUL / OP_ADD operand
	double c[5];                         // biquad c0..c4 (biquad.h:62-69 naming)
	double pad2;
	double P[CASCADE_NPOW][4];           // A^(2^k) row-major, A = [[-c3, 1], [-c4, 0]]  
	double h[CASCADE_L][2];              // first row of A^i, i = 0..L-1: zero-input response to the carried state
};

// Tables of the fast cascade kernel (kernels_cascade.hip, cascade_fast), per (channel, op), L = CASCADE_L:
//   fops [C][n_ops][FOP_DOUBLES]  wave-uniform constants, fetched with SCALAR loads (they live in SGPRs):
//        [0] kind (int64 bits)  [1] g  [2..6] c0..c4  [7] pad
//        [8..23] P^(L), P^(2L), P^(4L), P^(8L)  (row scan steps)   [24..27] P^(16L)  (row -> row carry)   [28..31] pad
//   fq   [C][n_ops][16][4]        Q[i] = P^(L (i + 1)): carry of the previous row's end state to lane i of a row
// with P = A = [[-c3, 1], [-c4, 0]].
constexpr int FOP_DOUBLES = 32, FOP_PW = 8, FOP_P16 = 24, FQ_DOUBLES = 64;
// cascade_rows: a wave = 4 channels x 16 lanes x ROWS_L consecutive frames; its tables are fops-shaped with L = ROWS_L
constexpr int ROWS_L = 32;

// The FFT convolver's input ring (one row per channel PAIR, 16-byte elements (x_a[n], x_b[n]) = the complex
// sequence the convolver transforms) that the cascade kernel may write instead of the interleaved block.
struct PlanarRing {
	double *base;                        // row r starts at base + 2 * r * row_stride (doubles)
	long row_stride;                     // elements (16 B) per row
	long mask;                           // ring length - 1 (power of two)
	long pos;                            // ring index of frame 0 of this block
	const int *pair_ch;                  // [rows_per_stream][2] channel feeding re / im (or -1: written as 0.0)
	int rows_per_stream;                 // pairs per stream
	int consecutive_pairs;               // pair q = channels (2q, 2q+1) for every q
};

struct CascadeParams {
	const double *in;                    // [S][frames][C]  (samples of in_fmt when that is not PCM_DOUBLE)
	double *out;                         // [S][out_stride][C] (may alias in)  (samples of sink.fmt when sink.on)
	int in_fmt;                          // PCM_DOUBLE, or a fusable wire format read by the kernel's own loads
	WireSink sink;
	long in_stride_frames, out_stride_frames;
	long frames;
	int C;                               // channels per stream
	int cg0, Cg;                         // channel group handled by blockIdx.y: [cg0 + y*Cg, ...)
	int n_ops;
	const OpDesc *ops;                   // [C][n_ops]
	const double *fops;                  // [C][n_ops][FOP_DOUBLES] fast-kernel constants (scalar loads)
	const double *fq;                    // [C][n_ops][FQ_DOUBLES] per-lane carry matrices
	const double *frows;                 // [C / 2][n_ops][FOP_DOUBLES] constants of cascade_rows (L = ROWS_L) per channel PAIR, or nullptr: the
	                                     // channels of a pair differ / ops other than sections and gains
	const double *frq;                   // [C / 2][n_ops][16][4] Q[i] = P^(ROWS_L (i + 1)) (cascade_rows<2>: lower row -> upper row carry)
	int rows4_ok;                        // the two pairs of every group of 4 channels are identical too (cascade_rows<4>)
	double *state;                       // [S][C][n_ops][2]
	PlanarRing ring;                     // optional second destination (ring.base != nullptr)
	int write_interleaved;               // 0: only the ring is written
	int xcd_map;                         // workgroup -> (stream, channel group) order that co-schedules a stream's groups on one XCD (set by launch_cascade)
};

// chunked cascade (kernels_chunk.hip): a call of K * len frames run as K zero-state chunks + carry + correction
struct ChunkParams {
	double *out;                         // [S][out_stride][C], holds the zero-state outputs
	long out_stride_frames, len;
	int C, K, D, n_pow, n_cls;           // D = 2 n_ops state variables per channel
	const int *cls;                      // [C] table index of the channel (channels with identical ops share tables)
	const double *H;                     // [n_cls][len][D] zero-input response at frame i to unit state k
	const double *Mp;                    // [n_cls][n_pow][D][D] M^(j+1), j = 0 .. n_pow-1 (n_pow = scan group size g); M = state transition over len frames
	const double *cstate;                // [S K][C][D] end states of the zero-state chunks
	double *X;                           // [S K][C][D] true state at the start of every chunk
	double *state;                       // [S][C][D] carried state: in = before the call, out = after it
};

// ---- pointwise kernels ----
struct RemixParams {
	const double *in; double *out;
	long in_stride_frames, out_stride_frames, frames;
	int Cin, Cout;
	const int *idx;                      // [Cout][max_n] input channel indices, ascending, -1 terminated rows
	int max_n;
	const double *w;                     // Mix: [Cout][max_n] weights (nullptr: plain remix sums from 0.0)
	const double *post;                  // Mix: [Cout] post-scale or nullptr
	int in_fmt;                          // `in` holds samples of this wire format (any of them; PCM_DOUBLE = plain)
	WireSink sink;                       // `out` receives the sink's output (any format)
};

struct DelayParams {                     // integer per-channel delay with carried ring (align.c:35-44)
	const double *in; double *out;
	long in_stride_frames, out_stride_frames, frames;
	int C;
	const long *len;                     // [C]
	const long *ring_off;                // [C] offset of the channel's ring in `ring` (per stream block of ring_per_stream)
	double *ring;                        // [S][ring_per_stream]
	long ring_per_stream;
	long pos;                            // frames already pushed through (same for all channels)
	int in_fmt;                          // as in RemixParams
	WireSink sink;
};

// ---- FFT convolution (overlap-save on channel pairs) ----
struct ConvGeom {
	int log2N, log2N1, log2N2;           // N = N1 * N2 complex points
	long N, N1, N2;
};

// Dynamic LDS above 64 KiB must be requested per kernel -- and the grant belongs to the DEVICE the attribute was set on.
// Thread-safe, keyed by (kernel, current device): chains live on several host threads (LADSPA multi-instance, the
// `watch` poller) and a process may drive several GPUs (engine.cpp).
void grant_dynamic_lds(const void *kernel, size_t bytes);

#ifdef __HIPCC__
// streaming (non-temporal) 16-byte accesses: data touched once per launch should not displace what the L2 keeps
typedef double dspamd_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 nt_load(const double2 *p) { const dspamd_v2d v = __builtin_nontemporal_load(reinterpret_cast<const dspamd_v2d *>(p)); return make_double2(v.x, v.y); }
__device__ __forceinline__ void nt_store(double2 *p, double2 a) { const dspamd_v2d v = { a.x, a.y }; __builtin_nontemporal_store(v, reinterpret_cast<dspamd_v2d *>(p)); }
__device__ __forceinline__ double2 ld16(const double2 *p, bool nt) { return nt ? nt_load(p) : *p; }
__device__ __forceinline__ void st16(double2 *p, double2 a, bool nt) { if (nt) nt_store(p, a); else *p = a; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global load and
// store of the wave (s_waitcnt vmcnt(0)), which would serialise prefetches and fire-and-forget stores with the
// compute phase they are meant to overlap.
__device__ __forceinline__ void lds_barrier()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
	__builtin_amdgcn_s_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#endif

}  // namespace dspamd
