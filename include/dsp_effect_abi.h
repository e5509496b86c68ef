/*
 * include/dsp_effect_abi.h -- the drop-in boundary, part 1: the plugin surface.
 *
 * libdsp_amd.so exports the reference's own effect entry points with the
 * reference's own signatures, so that the unmodified host (CLI dsp.c, LADSPA
 * frontend ladspa_dsp.c, the chain runtime effects_chain.c and the registry
 * effect.c:46-67) links against it instead of biquad.o / gain.o / remix.o /
 * delay.o / fir.o / fir_p.o / resample.o / hilbert.o / zita_convolver.o.
 * INTEGRATION.md shows the link line.
 *
 * The struct layouts below are ABI mirrors (field order, types) of:
 *   sample_t, struct stream_info ............ dsp.h:42, dsp.h:49-51
 *   struct effect_info ...................... effect.h:24-29
 *   EFFECT_FLAG_* ........................... effect.h:31-37
 *   struct effect (the vtable) .............. effect.h:39-59
 * A host translation unit compiled against the reference's effect.h and one
 * compiled against this header see the same objects.
 *
 * Every entry point takes plain C types only (no torch, no HIP types).
 */
#ifndef DSP_AMD_EFFECT_ABI_H
#define DSP_AMD_EFFECT_ABI_H

#include <sys/types.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef double sample_t;                    /* dsp.h:42 */

struct stream_info {                        /* dsp.h:49-51 */
	int fs, channels;
};

struct effect;

struct effect_info {                        /* effect.h:24-29 */
	const char *name;
	const char *usage;
	struct effect * (*init)(const struct effect_info *, const struct stream_info *, const char *, const char *, int, const char *const *);
	int effect_number;
};

enum {                                      /* effect.h:31-37 */
	EFFECT_FLAG_PLOT_MIX         = 1<<0,
	EFFECT_FLAG_OPT_REORDERABLE  = 1<<1,
	EFFECT_FLAG_NO_DITHER        = 1<<2,
	EFFECT_FLAG_CH_DEPS_IDENTITY = 1<<3,
	EFFECT_FLAG_ALIGN_BARRIER    = 1<<4,
};

struct effect {                             /* effect.h:39-59; "All functions may be NULL" */
	struct effect *prev, *next;
	const char *name;
	struct stream_info istream, ostream;
	char *channel_selector;
	int flags;
	int (*prepare)(struct effect *);
	sample_t * (*run)(struct effect *, ssize_t *, sample_t *, sample_t *);
	void (*reset)(struct effect *);
	void (*signal)(struct effect *);
	void (*plot)(struct effect *, int);
	void (*drain_samples)(struct effect *, ssize_t *);
	sample_t * (*drain2)(struct effect *, ssize_t *, sample_t *, sample_t *);
	void (*destroy)(struct effect *);
	int (*merge)(struct effect *, struct effect *);
	ssize_t (*buffer_frames)(struct effect *, ssize_t);
	void (*channel_deps)(struct effect *, char **);
	void (*channel_offsets)(struct effect *, ssize_t *, ssize_t *);
	void *data;
};

/*
 * Entry points (same six-argument init signature everywhere):
 *   (effect_info*, istream*, channel_selector [one byte per channel, util.h:48-53],
 *    dir [for relative filter paths], argc, argv [argv[0] = effect name])
 * Return: calloc'd struct effect (host frees it after ->destroy, effect.c:78-85),
 *         or NULL after logging to stderr.  An effect with run == NULL is a no-op
 *         that the host drops (effects_chain.c:586-590).
 */
struct effect *biquad_effect_init(const struct effect_info *, const struct stream_info *, const char *, const char *, int, const char *const *);        /* replaces biquad.h:74  */
struct effect *gain_effect_init(const struct effect_info *, const struct stream_info *, const char *, const char *, int, const char *const *);          /* replaces gain.h:31    */
struct effect *remix_effect_init(const struct effect_info *, const struct stream_info *, const char *, const char *, int, const char *const *);         /* replaces remix.h:25   */
struct effect *delay_effect_init(const struct effect_info *, const struct stream_info *, const char *, const char *, int, const char *const *);         /* replaces delay.h:27   */
struct effect *fir_effect_init(const struct effect_info *, const struct stream_info *, const char *, const char *, int, const char *const *);           /* replaces fir.h:28     */
struct effect *fir_p_effect_init(const struct effect_info *, const struct stream_info *, const char *, const char *, int, const char *const *);         /* replaces fir_p.h:28   */
struct effect *resample_effect_init(const struct effect_info *, const struct stream_info *, const char *, const char *, int, const char *const *);      /* replaces resample.h:26 */
struct effect *hilbert_effect_init(const struct effect_info *, const struct stream_info *, const char *, const char *, int, const char *const *);       /* replaces hilbert.h:26 */
struct effect *zita_convolver_effect_init(const struct effect_info *, const struct stream_info *, const char *, const char *, int, const char *const *); /* replaces zita_convolver.h:32 */
struct effect *st2ms_effect_init(const struct effect_info *, const struct stream_info *, const char *, const char *, int, const char *const *);         /* replaces st2ms.h:30 (st2ms, ms2st) */
struct effect *crossfeed_effect_init(const struct effect_info *, const struct stream_info *, const char *, const char *, int, const char *const *);     /* replaces crossfeed.h:25 */

/* filter-in-memory constructors used by other effects (hilbert.c:80,89; fir_p.c:365; matrix4_mb.c:776) */
struct effect *fir_effect_init_with_filter(const struct effect_info *, const struct stream_info *, const char *, sample_t *filter_data,
	int filter_channels, ssize_t filter_frames, ssize_t ref, int force_direct);                                  /* replaces fir.h:27   */
struct effect *fir_p_effect_init_with_filter(const struct effect_info *, const struct stream_info *, const char *, sample_t *filter_data,
	int filter_channels, ssize_t filter_frames, ssize_t ref, int max_part_len);                                  /* replaces fir_p.h:27 */
struct effect *zita_convolver_effect_init_with_filter(const struct effect_info *, const struct stream_info *, const char *, sample_t *filter_data,
	int filter_channels, ssize_t filter_frames, ssize_t ref, int min_part_len, int max_part_len);                /* replaces zita_convolver.h:31 */
struct effect *delay_effect_init_int(const char *name, const struct stream_info *, const char *, ssize_t samples_int);  /* replaces delay.h:25 */
struct effect *delay_effect_init_frac(const char *name, const struct stream_info *, const char *, double samples_frac, int fd_ap_n);  /* replaces delay.h:26 */

/* the effect numbers the registry passes in effect_info.effect_number */
enum {  /* biquad.h:30-51 */
	DSPAMD_BIQUAD_LOWPASS_1 = 1, DSPAMD_BIQUAD_HIGHPASS_1, DSPAMD_BIQUAD_ALLPASS_1, DSPAMD_BIQUAD_LOWSHELF_1,
	DSPAMD_BIQUAD_HIGHSHELF_1, DSPAMD_BIQUAD_LOWPASS_1P, DSPAMD_BIQUAD_LOWPASS, DSPAMD_BIQUAD_HIGHPASS,
	DSPAMD_BIQUAD_BANDPASS_SKIRT, DSPAMD_BIQUAD_BANDPASS_PEAK, DSPAMD_BIQUAD_NOTCH, DSPAMD_BIQUAD_ALLPASS,
	DSPAMD_BIQUAD_PEAK, DSPAMD_BIQUAD_LOWSHELF, DSPAMD_BIQUAD_HIGHSHELF, DSPAMD_BIQUAD_LOWPASS_TRANSFORM,
	DSPAMD_BIQUAD_HIGHPASS_TRANSFORM, DSPAMD_BIQUAD_DEEMPH, DSPAMD_BIQUAD_BIQUAD,
};
enum {  /* gain.h:25-29 */
	DSPAMD_GAIN_GAIN = 1, DSPAMD_GAIN_MULT, DSPAMD_GAIN_ADD,
};
enum {   /* effect_info.effect_number for st2ms_effect_init (st2ms.h:25-28) */
	DSPAMD_ST2MS_ST2MS = 1, DSPAMD_ST2MS_MS2ST,
};

/*
 * Optional import from the host ("symbols the shim may import"): when the process exports the
 * reference's fir_read_filter (fir_util.h:36, fir_util.c:25-120) -- i.e. when this library is linked into the reference
 * host -- filter files this library does not decode itself (anything but coefs: literals, raw PCM in double / float / s32 /
 * s24 / s16 and RIFF/WAVE) are read through it, i.e. through the host's own codec layer (init_codec, codec.h:72): every
 * container and encoding the host was built with.  Looked up with dlsym(RTLD_DEFAULT, "fir_read_filter"); absent (the
 * stand-alone hosts) such files are refused.  struct codec_params mirrors codec.h:57-60, the enums codec.h:24-34.
 */
struct dspamd_codec_params {
	const char *path, *type, *enc;
	int fs, channels, endian, mode, block_frames, buf_ratio;
};
enum { DSPAMD_CODEC_MODE_READ = 1 << 0 };
enum { DSPAMD_CODEC_ENDIAN_DEFAULT = 0, DSPAMD_CODEC_ENDIAN_BIG, DSPAMD_CODEC_ENDIAN_LITTLE, DSPAMD_CODEC_ENDIAN_NATIVE };
typedef sample_t *(*dspamd_host_fir_read_filter_fn)(const struct effect_info *, const struct stream_info *, const char *channel_selector,
                                                    const char *dir, const struct dspamd_codec_params *, int *channels, ssize_t *frames);

#ifdef __cplusplus
}
#endif
#endif
