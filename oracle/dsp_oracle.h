/*
 * oracle/dsp_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp64) of the reference's algorithms for the
 * effects_chain hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (dsp_amd/) never does.
 *
 * Pinning: the reference ships no golden vectors, so this
 * restatement is pinned against outputs of the reference itself: in-process
 * against oracle/_ref/libdspref.so (tests/test_oracle_vs_ref.py) and against
 * the committed fixtures in tests/golden/ that tests/golden/make_golden.py
 * generated from that same build of the reference.
 *
 * All objects are single-channel; callers loop over channels with a stride,
 * exactly as the reference's per-channel state arrays do.
 */
#ifndef DSP_ORACLE_H
#define DSP_ORACLE_H

#include <sys/types.h>

/* enum values follow biquad.h:30-60 */
enum {
	ORC_BIQUAD_LOWPASS_1 = 1, ORC_BIQUAD_HIGHPASS_1, ORC_BIQUAD_ALLPASS_1, ORC_BIQUAD_LOWSHELF_1,
	ORC_BIQUAD_HIGHSHELF_1, ORC_BIQUAD_LOWPASS_1P, ORC_BIQUAD_LOWPASS, ORC_BIQUAD_HIGHPASS,
	ORC_BIQUAD_BANDPASS_SKIRT, ORC_BIQUAD_BANDPASS_PEAK, ORC_BIQUAD_NOTCH, ORC_BIQUAD_ALLPASS,
	ORC_BIQUAD_PEAK, ORC_BIQUAD_LOWSHELF, ORC_BIQUAD_HIGHSHELF, ORC_BIQUAD_LOWPASS_TRANSFORM,
	ORC_BIQUAD_HIGHPASS_TRANSFORM,
};
enum { ORC_WIDTH_Q = 1, ORC_WIDTH_SLOPE, ORC_WIDTH_SLOPE_DB, ORC_WIDTH_BW_OCT, ORC_WIDTH_BW_HZ };

ssize_t orc_next_fast_fftw_len(ssize_t min_len);
double orc_parse_width(const char *s, int *type, int *ok);
void orc_biquad_coefs(double b0, double b1, double b2, double a0, double a1, double a2, double c[5]);
void orc_biquad_design(int type, double fs, double arg0, double arg1, double arg2, double arg3, int width_type, double c[5]);
void orc_biquad_run(const double c[5], double m[2], double *buf, ssize_t frames, int stride);

void orc_gain_run(double *buf, ssize_t frames, int channels, const double *mult);
void orc_add_run(double *buf, ssize_t frames, int channels, const double *add);
void orc_remix_run(const double *in, double *out, ssize_t frames, int in_channels, int out_channels, const char *sel);
void orc_delay_run(double *buf, ssize_t frames, int stride, double *ring, ssize_t len, ssize_t *p);
void orc_frac_delay_run(double *buf, ssize_t frames, int stride, int n, double delta, double *state);

void *orc_fir_direct_new(const double *taps, ssize_t n_taps);
void orc_fir_direct_run(void *st, double *buf, ssize_t frames, int stride);
void orc_fir_direct_free(void *st);

void *orc_fir_new(const double *taps, ssize_t n_taps);
ssize_t orc_fir_latency(void *st);
void orc_fir_run(void *st, double *buf, ssize_t frames, int stride);
void orc_fir_free(void *st);

int orc_fir_p_plan(ssize_t n_taps, int max_part_len, int single_thread, int len[4], int n[4], int delay[4]);
void *orc_fir_p_new(const double *taps, ssize_t n_taps, int max_part_len);
void orc_fir_p_run(void *st, double *buf, ssize_t frames, int stride);
void orc_fir_p_free(void *st);

void *orc_resample_new(int fs_in, int fs_out, double bw);
void orc_resample_params(void *st, int p[8]);  /* n, d, m, in_len, out_len, out_delay, sinc_fr_len, tmp_fr_len */
ssize_t orc_resample_run(void *st, const double *in, ssize_t frames, int istride, double *out, int ostride);
ssize_t orc_resample_drain(void *st, ssize_t frames, double *scratch_in, double *out);
void orc_resample_free(void *st);

void orc_hilbert_taps(ssize_t taps, double angle_deg, double *h);
void orc_sgen_sine(double *buf, ssize_t frames, int channels, int fs, double freq_hz, ssize_t pos0);
void orc_sgen_sweep(double *buf, ssize_t frames, int channels, int fs, double f0_hz, double f1_hz, ssize_t total_frames, ssize_t pos0);
void orc_sgen_delta(double *buf, ssize_t frames, int channels, ssize_t offset, ssize_t pos0);

/* fp64 direct-form linear convolution, full length n_x + n_taps - 1 (second oracle for fir/fir_p) */
void orc_conv_full(const double *x, ssize_t n_x, const double *taps, ssize_t n_taps, double *y);

void *orc_zita_equiv_new(const double *taps, ssize_t n_taps, int part_len);
void orc_zita_equiv_run(void *st, double *buf, ssize_t frames, int stride);
void orc_zita_equiv_free(void *st);

/* wire formats (DSPAMD_PCM_* numbering), clip and TPDF dither at the sink: sampleconv.c:25-149, dsp.c:673-699, util.h:127-178 */
void orc_pcm_read(int fmt, const void *in, double *out, ssize_t n);
void orc_pcm_write(int fmt, const double *in, void *out, ssize_t n, int dither_prec, unsigned int state[2], double stats[2]);

#endif
