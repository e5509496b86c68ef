/*
 * include/dsp_amd.h -- the drop-in boundary, part 2: stand-alone host and
 * device-resident batch entry points of libdsp_amd.so.
 *
 * Part 1 (dsp_effect_abi.h) is the plugin surface the reference's host binds.
 * This header is what a host WITHOUT the reference's chain runtime binds
 * (ctypes / cgo / JNI): the same life-cycle as effects_chain.h:40-53, plus a
 * batch axis ("S independent streams", which the reference can only express as
 * S processes) whose buffers live in HBM.
 *
 * All functions: plain C types, no torch / HIP types.  Device pointers are
 * passed as void* (hipMalloc'd or torch tensor.data_ptr()), streams as void*
 * (hipStream_t, e.g. torch.cuda.current_stream().cuda_stream; NULL = default).
 * Errors: functions returning pointers return NULL, functions returning
 * ssize_t/int return a negative value; dspamd_last_error() gives the message.
 * There is NO CPU fallback: without a usable HIP device every compute entry
 * point fails loudly.
 */
#ifndef DSP_AMD_H
#define DSP_AMD_H

#include <sys/types.h>
#include "dsp_effect_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *dspamd_version(void);
const char *dspamd_last_error(void);
/* number of visible HIP devices (0 if none); never touches oracle or CPU paths */
int dspamd_device_count(void);
/* bind the calling process to a device (one process per GPU); returns 0 on success */
int dspamd_set_device(int device);
/* LL_* of dsp.h:25-32 (0 silent .. 4 verbose); default 1 (errors) */
void dspamd_set_loglevel(int level);

/* registry lookup -- mirror of get_effect_info(), effect.c:69-76, for the effects this library provides */
const struct effect_info *dspamd_get_effect_info(const char *name);

/* host-side planning only (no device needed): build the chain plan -- parse, merge (effects_chain.c:605-641), prepare
 * (:925-932), alignment, drain accounting -- and copy the FIR that effect number `effect` of the resulting chain
 * applies to `channel`: the taps of fir / fir_p / hilbert (fir_util.c:25-120, hilbert.c:28-92), or the FIR a
 * time-reversed IIR effect (`biquad -r`, reverse_iir.c:381-636) is designed into.  Returns the tap count (may exceed
 * max_taps), 0 if the channel is not filtered, -1 on error; *delay = the delay reported through channel_offsets. */
ssize_t dspamd_plan_fir(const char *chain_str, int fs, int channels, const char *dir, int effect, int channel, double *taps, ssize_t max_taps, ssize_t *delay);

/* ---- stand-alone chain on HOST buffers (one stream; mirror of effects_chain.h:40-53) ---- */
typedef struct dspamd_chain dspamd_chain;

/* build_effects_chain_from_string() + optimise/prepare/align/drain accounting (effects_chain.c:934-969) */
dspamd_chain *dspamd_chain_build(const char *chain_str, int fs, int channels, const char *dir, int *out_fs, int *out_channels);
/* run_effects_chain(), effects_chain.c:1058: interleaved fp64 frames in, frames out (may differ); returns frames written or <0 */
ssize_t dspamd_chain_run(dspamd_chain *, const double *in, ssize_t frames, double *out, ssize_t out_capacity_frames);
/* drain_effects_chain(), effects_chain.c:1186: returns frames written, or -1 when dry */
ssize_t dspamd_chain_drain(dspamd_chain *, ssize_t block_frames, double *out, ssize_t out_capacity_frames);
/* get_effects_chain_max_out_frames(), effects_chain.c:1015-1020 */
ssize_t dspamd_chain_max_out_frames(dspamd_chain *, ssize_t in_frames);
ssize_t dspamd_chain_drain_frames(dspamd_chain *);   /* chain->drain_frames, effects_chain.c:877-923 */
void dspamd_chain_reset(dspamd_chain *);             /* reset_effects_chain(), effects_chain.c:1091 */
void dspamd_chain_destroy(dspamd_chain *);           /* destroy_effects_chain(), effects_chain.c:1220 */
int dspamd_chain_n_effects(dspamd_chain *);
const char *dspamd_chain_effect_name(dspamd_chain *, int i);

/* ---- device-resident batch: S independent streams x C channels, buffers in HBM ---- */
typedef struct dspamd_batch dspamd_batch;

/*
 * Every stream runs the same chain (filters are shared, state is per stream).
 * Layout of every device buffer: [stream][frame][channel] fp64, i.e. each
 * stream's slab is exactly what the reference's run() would be handed
 * (dsp.h:42, effects_chain.c:1044-1056).
 * max_frames: the largest `frames` that will be passed to dspamd_batch_run.
 */
dspamd_batch *dspamd_batch_create(const char *chain_str, int fs, int channels, int n_streams, ssize_t max_frames, const char *dir);
int dspamd_batch_out_fs(dspamd_batch *);
int dspamd_batch_out_channels(dspamd_batch *);
ssize_t dspamd_batch_max_out_frames(dspamd_batch *, ssize_t in_frames);
ssize_t dspamd_batch_drain_frames(dspamd_batch *);
/*
 * One block for all streams, asynchronously on `stream`; no host sync, no
 * allocation (after the first call of a size: plans and LDS grants are made on
 * first use), capturable into a hipGraph.  A captured call carries its position
 * in the streams (ring offsets of convolvers, delays and resamplers are launch
 * parameters): replaying it continues the streams only for chains whose state
 * lives entirely in device memory (gains, biquad sections, remix); for the others
 * capture a whole period of calls, or re-capture per call.  d_in: [S][frames][C_in]; d_out: [S][out_stride_frames][C_out]
 * (out_stride_frames >= returned frame count; pass 0 for "= max_out_frames(frames)").
 * Returns frames produced per stream, or <0.
 */
ssize_t dspamd_batch_run(dspamd_batch *, const void *d_in, ssize_t frames, void *d_out, ssize_t out_stride_frames, void *stream);
/* the same with the streams' input slabs in_stride_frames apart (>= frames): d_in is [S][in_stride_frames][C_in].  Slabs
 * whose distance is a large power of two (or a multiple of one: 196608 frames x 64 B = 3 x 4 MiB) put the same frame of every
 * stream on the same memory channels; a few hundred bytes of padding per stream avoid that. */
ssize_t dspamd_batch_run_strided(dspamd_batch *, const void *d_in, ssize_t in_stride_frames, ssize_t frames, void *d_out, ssize_t out_stride_frames, void *stream);
/* end of stream: push zeros / flush rate changers; returns frames produced, -1 when dry */
ssize_t dspamd_batch_drain(dspamd_batch *, ssize_t block_frames, void *d_out, ssize_t out_stride_frames, void *stream);
void dspamd_batch_reset(dspamd_batch *, void *stream);
void dspamd_batch_destroy(dspamd_batch *);
/* introspection for tests / DESIGN.md: human-readable stage plan ("cascade[gain+10 biquad] -> conv[N=131072 ...]") */
const char *dspamd_batch_plan(dspamd_batch *);
/* names + stream of the kernels the batch launches per run (for HIP-event timing on the right stream) */
int dspamd_batch_n_stages(dspamd_batch *);

/* ---- per-kernel timing with HIP events on the launch stream (bench.py "roofline") ---- */
void dspamd_profile_enable(int on);
/* device-synchronises; text lines "kernel_name total_ms launches" for everything launched since the last collect */
const char *dspamd_profile_collect(void);

/* ---- device-side bench endpoints (sgen.c:55-67 / null.c:31-34 equivalents) ---- */
/* stream s, frame t (absolute position pos0+t), every channel: sin(2 pi (freq0 + s*dfreq) * (pos0+t)/fs) */
int dspamd_sgen_sine(void *d_buf, int n_streams, ssize_t frames, int channels, int fs, double freq0, double dfreq, ssize_t pos0, void *stream);
/* the generator's other two forms: `sine:freq=f0-f1` on a source of total_frames (sgen.c:60-62, 163: exponential sweep) and `delta:offset=N`
 * (sgen.c:46-52); stream s uses freq0 + s dfreq with the same ratio freq1 / freq0, resp. offset + s doffset; pos0 = frames already generated */
int dspamd_sgen_sweep(void *d_buf, int n_streams, ssize_t frames, int channels, int fs, double freq0, double freq1, double dfreq, ssize_t total_frames, ssize_t pos0, void *stream);
int dspamd_sgen_delta(void *d_buf, int n_streams, ssize_t frames, int channels, ssize_t offset, ssize_t doffset, ssize_t pos0, void *stream);
/* per-stream digests like stats.c:47-76: sum, sum of squares, peak -> d_out[S][3] */
int dspamd_digest(const void *d_buf, int n_streams, ssize_t frames, ssize_t stride_frames, int channels, void *d_out, void *stream);
/* ---- wire formats on the device (sampleconv.c:25-149 with the BIT_PERFECT macros of sampleconv.h:35-56) ---- */
enum { DSPAMD_PCM_U8 = 0, DSPAMD_PCM_S8, DSPAMD_PCM_S16, DSPAMD_PCM_S24 /* in 32 bits */, DSPAMD_PCM_S32, DSPAMD_PCM_S24_3 /* packed */,
       DSPAMD_PCM_FLOAT, DSPAMD_PCM_DOUBLE };
size_t dspamd_pcm_sample_bytes(int fmt);
/* read_buf_<fmt>: n_samples of the wire format -> fp64 samples */
int dspamd_pcm_read(int fmt, const void *d_in, void *d_out, ssize_t n_samples, void *stream);
/* the output stage of dsp.c:673-694: [TPDF dither at dither_prec bits (0 = none; util.h:127-178: two Lehmer generators seeded
 * with 1, advanced once per sample in interleaved order, frames_before = frames of each stream already written)], clip() to
 * [-1, 1], write_buf_<fmt>.  in: [S][in_stride_frames][channels] fp64; out: packed [S][frames][channels] of the format;
 * d_stats (optional, zeroed by the caller): [S][2] = clipped samples (64-bit count), peak |sample| (fp64) */
int dspamd_pcm_write(int fmt, const void *d_in, ssize_t in_stride_frames, void *d_out, int n_streams, ssize_t frames, int channels,
                     int dither_prec, ssize_t frames_before, void *d_stats, void *stream);

/*
 * One block from wire format to wire format -- the file -> file path of dsp.c: read_buf_<in_fmt>, the chain, then the sink
 * above at the position the batch has reached (frames written through run_wire / drain_wire since create / reset).
 * d_in: [S][in_stride_frames][C_in] samples of in_fmt (0 = frames); d_out: [S][out_stride_frames][C_out] samples of out_fmt
 * (0 = max_out_frames(frames)); d_stats as for dspamd_pcm_write.  Where the first / last kernel of the plan can, it converts in
 * its own loads / stores -- no separate passes over the block: the biquad cascade kernel on either side (512 channels and more,
 * identical sections on the channels of a group), a convolver's first kernel on the input side, the inverse column transform of
 * a plain convolution or of a 2x upsampler on the output side (s16 / s24 / s32 / float / double each), remix and the alignment
 * delay on either side (every format) -- otherwise the conversion kernels run before / after it on buffers of the batch
 * (docs/history.md section 4.6 has the table).  The samples are the same either way, bit for bit; dspamd_batch_wire_fused() says
 * what the last call did.
 */
ssize_t dspamd_batch_run_wire(dspamd_batch *, int in_fmt, const void *d_in, ssize_t in_stride_frames, ssize_t frames,
                              int out_fmt, void *d_out, ssize_t out_stride_frames, int dither_prec, void *d_stats, void *stream);
ssize_t dspamd_batch_drain_wire(dspamd_batch *, ssize_t block_frames, int out_fmt, void *d_out, ssize_t out_stride_frames, int dither_prec, void *d_stats, void *stream);
/* what the last run_wire / drain_wire did: bit 0 = input format read by the first kernel, bit 1 = sink applied by the last kernel */
int dspamd_batch_wire_fused(dspamd_batch *);

/* How the plugin path (effect->run() on host buffers, effect.h:47) has served its blocks in this process, for tests and field reports:
 * out[0] blocks served by a resident wave (no launch: kernels_resident.hip), [1] small blocks through the mapped staging buffers and a launch,
 * [2] larger blocks (copies + launches), [3] resident kernels started, [4] blocks a wave did not serve in time, [5] segments whose resident
 * path was switched off.  Returns the number of values written (<= n). */
int dspamd_plugin_counters(long long *out, int n);

/* plain device copy kernel: measured HBM ceiling next to the 8 TB/s spec (bytes must be a multiple of 16) */
int dspamd_copy_probe(const void *d_src, void *d_dst, size_t bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
