// kparams.h -- plain-data launch parameters shared by the HIP kernels and the host engine.
#pragma once
#include <cstddef>
#include <cstdint>
#include <sys/types.h>

namespace dspamd {

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

// The FFT convolver's input ring (one row per channel PAIR, 16-byte elements (x_a[n], x_b[n]) = the complex
// sequence the convolver transforms) that the cascade kernel may write instead of the interleaved block.
struct PlanarRing {
	double *base;                        // row r starts at base + 2 * r * row_stride (doubles)
	long row_stride;                     // elements (16 B) per row
	long mask;                           // ring length - 1 (power of two)
	long pos;                            // ring index of frame 0 of this block
	const int *pair_ch;                  // [rows_per_stream][2] channel feeding re / im (or -1: written as 0.0)
	int rows_per_stream;                 // pairs per stream
};

struct CascadeParams {
	const double *in;                    // [S][frames][C]
	double *out;                         // [S][out_stride][C] (may alias in)
	long in_stride_frames, out_stride_frames;
	long frames;
	int C;                               // channels per stream
	int cg0, Cg;                         // channel group handled by blockIdx.y: [cg0 + y*Cg, ...)
	int n_ops;
	const OpDesc *ops;                   // [C][n_ops]
	double *state;                       // [S][C][n_ops][2]
	PlanarRing ring;                     // optional second destination (ring.base != nullptr)
	int write_interleaved;               // 0: only the ring is written
};

// ---- pointwise kernels ----
struct RemixParams {
	const double *in; double *out;
	long in_stride_frames, out_stride_frames, frames;
	int Cin, Cout;
	const int *idx;                      // [Cout][max_n] input channel indices, ascending, -1 terminated rows
	int max_n;
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
};

// ---- FFT convolution (overlap-save on channel pairs) ----
struct ConvGeom {
	int log2N, log2N1, log2N2;           // N = N1 * N2 complex points
	long N, N1, N2;
};

}  // namespace dspamd
