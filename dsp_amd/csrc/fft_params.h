// fft_params.h -- launch parameters of the FFT convolver kernels (kernels_fft.hip) shared with conv.cpp
#pragma once
#include <hip/hip_runtime.h>
#include "kparams.h"

namespace dspamd {

constexpr int FFT_MIN_LOG2_N2 = 9, FFT_MAX_LOG2_N2 = 12;   // contiguous (row) dimension: 512 .. 4096 points = 16 * 16 * {2, 4, 8, 16}
constexpr int FFT_MIN_LOG2_N1 = 4, FFT_MAX_LOG2_N1 = 8;    // strided (column) dimension: 16 .. 256 points = 16 * {1, 2, 4, 8, 16}
constexpr int FIR_DIRECT_MAX = 32;      // fir_p.c:34 DIRECT_LEN

struct ConvParams {
	int log2N1, log2N2, log2_lo;
	long N, N1, N2;
	// input window of pair r: z[n] = ring[r * ring_row_stride + ((win_base + n) & ring_mask)] for n < valid, else 0.
	// The ring holds the complex sequence itself: (first channel of the pair, second channel or 0.0) per frame.
	const double2 *ring;
	long ring_row_stride, ring_mask, win_base, valid;
	// K1 direct mode (slab != nullptr): every window element that lies in the CURRENT call's input -- slab frame
	// fr = slab_frame0 + n - first_n >= 0 of stream pair / pairs_per_stream, channels (2 q, 2 q + 1) -- comes straight from
	// the interleaved slab, older ones from the ring; with slab_store K1 itself files the frames that later windows need
	// (the last first_n of the block) in the ring
	const double *slab;
	long slab_stride_frames, slab_frame0;   // slab frame of window element first_n (negative: the window starts in older calls)
	int slab_store;                         // 1: K1 files the history itself (plain convolution); 0: the host pushes the tail of the call
	int slab_fmt;                           // PCM_DOUBLE, or the fusable wire format the slab holds (first kernel of a pipeline run in wire formats)
	const int *pair_h;                  // [n_pairs] index of the filter spectrum used by the pair
	int shared_h;                       // every pair uses filter 0 (pair_h is all zeros)
	long pair0;                         // first pair handled by this launch (W is indexed relative to it)
	double2 *W;                         // [pairs in chunk][w_stride] work spectrum / time buffer
	long w_stride;                      // N + a few KB: K3 reads the 4 pairs of a stream at once, and at a power-of-two distance they
	                                    // queue up on the same memory channels (scripts/ubench/hbmprobe.hip: 4.7 -> 5.1 TB/s for its pattern)
	const double2 *tw_n1, *tw_n2;       // exp(-2 pi i k / N1), exp(-2 pi i k / N2)
	const double2 *tw_hi, *tw_lo;       // w_N^(hi << log2_lo), w_N^lo
	const double2 *tw_col;              // [N1 / 16 + 1][N2]: w_N^(n2 j) for j < N1 / 16, then w_N^(n2 N1 / 16): the inter-pass twiddle as K1 / K3 read it
	const double2 *H;                   // [n_filters][N] filter spectra in [k1][k2] order, pre-scaled by 1/N
	double2 *Hout;                      // mode 1 of conv_row
	double h_scale;
	// output (K3).  Window sample first_n + f (0 <= f < in_count) is the convolution at input index q_blk + f; phase ph
	// of it is output index K = up * (q_blk + f) + ph, written -- when down divides K -- to frame K / down - k_origin
	// (if inside [0, out_count)).  Plain convolution: nph = up = down = 1, q_blk = 0, k_origin = -(first frame of the block).
	double *out;
	long out_stride_frames, first_n, in_count, out_count, q_blk, k_origin;
	int nph, up, down;                  // nph phases: filter spectrum (pair_h * nph + ph), W + ph * phase_stride
	long phase_stride;
	// optional: the NEXT convolver's pair ring as destination instead of the interleaved slab (same pairing of
	// channels): element (pair q of stream s, frame mo) at ring_out[(s * pps + q) * ring_out_stride + ((ring_out_pos + mo) & ring_out_mask)]
	double2 *ring_out;
	long ring_out_stride, ring_out_mask, ring_out_pos;
	int ring_out_round_f32;
	int nt;                             // non-temporal access mask: 1 K1 ring loads, 2 K1 W stores, 4 K2 W loads, 8 K2 W stores, 16 K3 W loads, 32 K3 out stores
	int C, pairs_per_stream;
	long stream0, n_streams_launch;
	const int *pair_out_ch;             // [pairs_per_stream][2] channel written by re / im (or -1)
	int round_f32;
	int k3_pipe_ok;                     // every stream is four pairs of adjacent channels of an 8-channel, 16-byte aligned slab: K3 may take its persistent form
	// mode 3 of conv_row (uniformly partitioned convolution through the four-step transform): the pairs' delay lines
	// fdl[slot][pair][N] (slot stride in elements), fdl_P partitions whose spectra are H + q N, fdl_slot = the slot this launch writes
	double2 *fdl;
	long fdl_slot_stride;
	int fdl_P, fdl_slot;
	int f32;                            // 1: W, H and the twiddle tables hold float2 (the float32 instance, kernels_fft32.hip); rings / slabs / outputs stay fp64
	WireSink sink;                      // K3 of a plain convolution at the end of a pipeline: `out` holds samples of sink.fmt (kparams.h)
};

// Small-call regime (calls much shorter than the filter): the head of the filter as a uniformly partitioned convolution with a
// frequency-domain delay line -- what fft_part_group_compute does on the CPU (fir_p.c:64-103) -- with the partition = the
// call's block B, transform size NF = 2 B; the rest of the filter is served by the overlap-save convolver once per P1 B frames.
struct FdlParams {
	int log2NF, P1;                     // NF = 2 B points per transform (512 .. 4096), P1 partitions of B taps
	long NF, B;
	const double2 *ring;                // the pair rings (complex samples), as in ConvParams
	long ring_row_stride, ring_mask;
	long win_base;                      // ring index of the first window element of sub-block 0 (= first new frame - B)
	int n_sub;                          // sub-blocks of B frames handled by this launch, one after the other
	int slot0;                          // delay-line slot written by sub-block 0; sub-block b writes (slot0 + b) % P1
	double2 *fdl;                       // [P1][n_pairs][NF] spectra of the last P1 windows, in the transform's own (natural) order
	const double2 *Hf;                  // [n_filters][P1][NF] spectra of the zero-padded head partitions, pre-scaled by 1 / NF
	const int *pair_h;                  // [n_pairs] filter of the pair (one filter per channel, fir_p.c:483-495), or nullptr: every pair uses filter 0
	double2 *spec_out;                  // preparation mode: write h_scale x forward transform here ([n_pairs][NF]) and stop
	double h_scale;
	const double2 *tw_nf;               // exp(-2 pi i k / NF)
	long n_pairs;
	int C, pairs_per_stream;
	const int *pair_out_ch;             // [pairs_per_stream][2]
	double *out;                        // [S][out_stride_frames][C]
	long out_stride_frames;
	const double *tail;                 // [S][tail_stride_frames][C]: contribution of the taps from P1 B on, or nullptr
	long tail_stride_frames, tail_off;  // frame tail_off + f of `tail` belongs to output frame f of this launch
	WireSink sink;                      // the stage is the last of a pipeline run in wire formats: `out` holds samples of sink.fmt (kparams.h)
	int round_f32;                      // the zita contract: outputs (head + tail share) rounded to float32 (zita_convolver.cpp:110)
};
void launch_conv_fdl(const FdlParams &p, hipStream_t st);

struct DeintParams {
	const double *in;
	double *out;                        // pass-through destination for unselected channels (may be null)
	long in_stride_frames, out_stride_frames, frames;
	int C;
	const int *slot_of_channel;         // [C] 2 * (pair within the stream) + (0 = re, 1 = im), or -1 (not convolved)
	int rows_per_stream;                // pairs per stream
	double2 *ring;
	long ring_row_stride, ring_mask, pos;
	int round_f32;
	int in_fmt;                         // PCM_DOUBLE, or the wire format `in` holds
};

struct FirDirectParams {
	const double *in;
	double *out;
	long in_stride_frames, out_stride_frames, frames;
	int C, T;
	const int *filter_of_channel;       // [C] filter index or -1 (pass through)
	const double *taps;                 // [n_filters][FIR_DIRECT_MAX]
	const double *hist_rd;              // [S][C][FIR_DIRECT_MAX]  x[-(q+1)]
	double *hist_wr;
};

// The cascade fused into the convolver's first pass (round 4, kernels_fused.hip): a call of exactly one hop whose window rows
// are whole chunks of the recurrence.  The window of N = N1 x N2 points is [hist_rows rows of history | N1 - hist_rows rows of new
// frames]; row r (N2 consecutive frames) of every channel is cut into `seg` chunks; chunk c = (r - hist_rows) seg + sg.
//   fused_prepass    every chunk run from ZERO state, its end state -> cstate           (reads the slab once, writes K D per channel)
//   cascade_chunk_carry (kernels_chunk.hip)   the scan over the chunks with M = A^len   -> X = true state at every chunk start
//   fused_col_fwd    K1 with the recurrence in front of its column transforms: the slab's frames -> sections (from X) -> W;
//                    the last hist_rows rows of cascade output go to the pair rings on the way (the next window's history)
// so the ring round trip of the cascade's output (write + K1's read: a quarter of the step's HBM traffic) is gone.
struct FuseParams {
	const double *in;                   // [S][in_stride_frames][C] slab (fp64, or samples of in_fmt); frame 0 = window element first_n
	int in_fmt;                         // PCM_DOUBLE, or the fusable wire format the slab holds (first stage of a pipeline run in wire formats: read_buf_<fmt>
	                                    // in the loads of the matrix-core prepass and of the fused first pass)
	long in_stride_frames;
	int C, n_sec, n_ops;                // n_sec biquad sections (gains folded in), n_ops ops per channel in the state layout (D = 2 n_ops)
	const int *sec_op;                  // [n_sec] op index whose (m0, m1) the section carries
	double gain;                        // product of the gains behind the last section
	long sec_stride;                    // 0: one section table for every pair; else doubles from one pair's table [n_sec][6] to the next pair's (pair = channels 2 q, 2 q + 1 of a stream)
	const double *gain_tab;             // [pairs per stream] the gains behind the last section, per pair (sec_stride != 0), or nullptr
	int seg, hist_rows;
	long K, len;                        // chunks per channel = (N1 - hist_rows) seg, frames per chunk = N2 / seg
	double *cstate;                     // [S K][C][D]
	const double *X;                    // [S K][C][D]
	int n_streams;
};
constexpr int FUSE_MAX_SEC = 12;
// sec: [n_sec][6] device table c0 c1 c2 c3 c4 pad (read with scalar loads); false: no instance for this section count
bool launch_fused_prepass(const FuseParams &f, const double *sec, long N2, int pps, hipStream_t st);
bool launch_fused_prepass_mm(const FuseParams &f, const double *Gt, long N2, int n_state, hipStream_t st);   // false: shape not served, take the other one
int fused_section_slots(int n_sec);   // section count of the instance that takes n_sec sections (the host pads with pass-through sections), 0 = none
bool launch_fused_col_fwd(const ConvParams &p, const FuseParams &f, const double *sec, hipStream_t st);

// One-trip convolver for SHORT filters behind long calls (round 5, kernels_short.hip): taps - 1 <= N / 2, N = 8192: a pair's whole transform lives in one
// workgroup's LDS, so a block of hop = N - first_n frames is one read of the window and one write of the outputs -- no W, no second and third trip.
// Window element n of block b is input index q0 + b hop - lat - first_n + n (ring position = index & ring_mask; from the slab instead where the
// index lies in the current call and the stage reads its slab directly); elements at or beyond first_n + in_count are zero; window sample
// first_n + f is the convolution at input index q0 + b hop + f, written to output frame q0 + b hop + f - k_origin when inside [0, out_count).
struct ShortParams {
	long N, first_n, hop;
	const double2 *ring;
	long ring_row_stride, ring_mask;
	long q0, lat, n_in;                 // first input index of block 0, the stage's latency, input indices covered (blocks of hop frames, the last one shorter)
	const double *slab;                 // direct mode: [S][slab_stride_frames][C], frame 0 = input index slab_q0; nullptr = everything from the rings
	long slab_stride_frames, slab_q0, file_from;   // slab frames with input index >= file_from are filed in the ring on the way (the history later calls look back at)
	long slab_frames;                   // frames of the current call in every stream's slab (what the slab descriptor lets the kernel read)
	int slab_fmt;                       // PCM_DOUBLE, or the fusable wire format the slab holds (first kernel of a pipeline run in wire formats; the rings keep fp64)
	WireSink sink;                      // last kernel of such a pipeline: `out` holds samples of sink.fmt (dither, clip, conversion in the stores)
	int C, pairs_per_stream;
	const int *pair_out_ch;             // [pairs_per_stream][2]
	const int *pair_h;                  // [n_pairs] filter of the pair
	const double2 *H;                   // [n_filters][N] (fdl_P > 0: [n_filters][fdl_P][N]), natural order, pre-scaled by 1 / N
	// uniformly partitioned form (as ConvParams' fdl fields): fdl[slot][pair][N] (slot stride in elements), fdl_P partitions of hop taps, fdl_slot = the slot block 0 of
	// this launch writes (block b: (fdl_slot + b) % fdl_P); the launch then runs its blocks in order, one workgroup per pair
	double2 *fdl;
	long fdl_slot_stride;
	int fdl_P, fdl_slot;
	double2 *Hout;                      // preparation mode: h_scale x forward transform of the window goes here ([n_pairs][N]) and the kernel stops
	double h_scale;
	const double2 *tw;                  // exp(-2 pi i k / N), k < N
	double *out;
	long out_stride_frames, k_origin, out_count;
	double2 *ring_out;                  // the NEXT convolver's pair rings as destination (as in ConvParams), or nullptr
	long ring_out_stride, ring_out_mask, ring_out_pos;
	int ring_out_round_f32, round_f32;
	long n_pairs;
	int blocks_per_wg;
};
void launch_conv_short(const ShortParams &p, hipStream_t st);
constexpr long CONV_SHORT_N = 8192, CONV_SHORT_N2 = 16384;   // the two window sizes of the one-trip convolver (kernels_short.hip)

void launch_conv_col(const ConvParams &p, bool inverse, int grid_y, hipStream_t st);
void launch_conv_row(const ConvParams &p, int mode, int n_pairs, hipStream_t st);
void launch_deinterleave(const DeintParams &p, int n_streams, hipStream_t st);
void launch_fir_direct(const FirDirectParams &p, int n_streams, hipStream_t st);

}  // namespace dspamd
