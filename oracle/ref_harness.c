/*
 * oracle/ref_harness.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * In-process embedding of the *real* reference (objects compiled by
 * oracle/Makefile straight from /root/reference, never copied into this repo)
 * for three purposes:
 *   1. pinning oracle/dsp_oracle.c (the restatement) against the reference,
 *   2. generating the golden vectors under tests/golden/,
 *   3. the "reference" CPU baseline that bench.py times next to the GPU path.
 *
 * The embedding recipe is the one the reference's own LADSPA frontend uses
 * (ladspa_dsp.c:55-73 defines dsp_globals and the log lock; dsp.h:53-71 lists
 * the host hooks): link every reference object except dsp.o, provide the eight
 * host hooks, then drive effects_chain.h:40-53
 * (build_effects_chain_from_string / run_effects_chain / drain_effects_chain).
 *
 * Everything exported here is prefixed refh_ and uses plain C types so that it
 * can be driven through ctypes.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "dsp.h"
#include "effect.h"
#include "effects_chain.h"
#include "util.h"

struct dsp_globals dsp_globals = {
	LL_ERROR,      /* loglevel */
	"dsp_ref",     /* prog_name */
};

static pthread_mutex_t log_lock = PTHREAD_MUTEX_INITIALIZER;
void dsp_log_acquire(void) { pthread_mutex_lock(&log_lock); }
void dsp_log_release(void) { pthread_mutex_unlock(&log_lock); }
void dsp_statuslines_acquire(void) {}
void dsp_statuslines_release(void) {}
void dsp_statusline_register(struct statusline_state *s) { (void) s; }
void dsp_statusline_unregister(struct statusline_state *s) { (void) s; }
void dsp_get_term_size(int *w, int *h) { if (w) *w = 80; if (h) *h = 24; }

struct refh_chain {
	struct effects_chain chain;
	struct stream_info istream, ostream;
	sample_t *buf1, *buf2;
	ssize_t buf_frames;  /* input frames the buffers were sized for */
};

void refh_set_loglevel(int l) { dsp_globals.loglevel = l; }

static int refh_realloc(struct refh_chain *h, ssize_t in_frames)
{
	if (in_frames <= h->buf_frames) return 0;
	const ssize_t len = get_effects_chain_buffer_len(&h->chain, in_frames, h->istream.channels);
	free(h->buf1);
	free(h->buf2);
	h->buf1 = calloc(len, sizeof(sample_t));
	h->buf2 = calloc(len, sizeof(sample_t));
	if (!h->buf1 || !h->buf2) return 1;
	h->buf_frames = in_frames;
	return 0;
}

/* dir: directory against which relative filter paths resolve (may be NULL) */
void *refh_chain_new(const char *chain_str, int fs, int channels, const char *dir, int *out_fs, int *out_channels)
{
	struct refh_chain *h = calloc(1, sizeof(*h));
	if (!h) return NULL;
	struct stream_info stream = { .fs = fs, .channels = channels };
	h->istream = stream;
	if (build_effects_chain_from_string(chain_str, NULL, &h->chain, &stream, NULL, dir)) {
		destroy_effects_chain(&h->chain);
		free(h);
		return NULL;
	}
	h->ostream = stream;
	if (out_fs) *out_fs = stream.fs;
	if (out_channels) *out_channels = stream.channels;
	return h;
}

void refh_chain_free(void *hp)
{
	struct refh_chain *h = hp;
	if (!h) return;
	destroy_effects_chain(&h->chain);
	free(h->buf1);
	free(h->buf2);
	free(h);
}

void refh_chain_reset(void *hp)
{
	struct refh_chain *h = hp;
	reset_effects_chain(&h->chain);
}

ssize_t refh_chain_max_out_frames(void *hp, ssize_t in_frames)
{
	struct refh_chain *h = hp;
	return get_effects_chain_max_out_frames(&h->chain, in_frames);
}

ssize_t refh_chain_drain_frames(void *hp)
{
	struct refh_chain *h = hp;
	return h->chain.drain_frames;
}

int refh_chain_n_effects(void *hp)
{
	struct refh_chain *h = hp;
	int n = 0;
	for (struct effect *e = h->chain.head; e; e = e->next) ++n;
	return n;
}

const char *refh_chain_effect_name(void *hp, int i)
{
	struct refh_chain *h = hp;
	struct effect *e = h->chain.head;
	while (e && i-- > 0) e = e->next;
	return e ? e->name : NULL;
}

/* one run_effects_chain() call (effects_chain.c:1058); returns frames produced */
ssize_t refh_chain_run(void *hp, const double *in, ssize_t frames, double *out)
{
	struct refh_chain *h = hp;
	if (frames < 1) return 0;
	if (refh_realloc(h, frames)) return -1;
	memcpy(h->buf1, in, (size_t) frames * h->istream.channels * sizeof(sample_t));
	ssize_t f = frames;
	sample_t *r = run_effects_chain(&h->chain, &f, h->buf1, h->buf2);
	if (f > 0) memcpy(out, r, (size_t) f * h->ostream.channels * sizeof(sample_t));
	return f;
}

/* one drain_effects_chain() call (effects_chain.c:1186); returns -1 when dry */
ssize_t refh_chain_drain(void *hp, ssize_t block_frames, double *out)
{
	struct refh_chain *h = hp;
	if (refh_realloc(h, block_frames)) return -1;
	ssize_t f = block_frames;
	sample_t *r = drain_effects_chain(&h->chain, &f, h->buf1, h->buf2);
	if (f > 0) memcpy(out, r, (size_t) f * h->ostream.channels * sizeof(sample_t));
	return f;
}

/*
 * Whole-stream convenience: feed `frames` input frames in blocks of
 * `block_frames`, then drain, exactly like the CLI loop (dsp.c:1295-1454).
 * Returns the number of output frames written (<= out_cap).
 */
ssize_t refh_chain_process(void *hp, const double *in, ssize_t frames, ssize_t block_frames, double *out, ssize_t out_cap)
{
	struct refh_chain *h = hp;
	const int ic = h->istream.channels, oc = h->ostream.channels;
	ssize_t pos = 0, opos = 0;
	const ssize_t max_of = get_effects_chain_max_out_frames(&h->chain, block_frames);
	sample_t *tmp = malloc((size_t) (max_of > block_frames ? max_of : block_frames) * (oc > ic ? oc : ic) * sizeof(sample_t));
	if (!tmp) return -1;
	while (pos < frames) {
		const ssize_t n = (frames - pos < block_frames) ? frames - pos : block_frames;
		const ssize_t f = refh_chain_run(h, in + pos * ic, n, tmp);
		if (f < 0) { free(tmp); return -1; }
		const ssize_t c = (opos + f <= out_cap) ? f : out_cap - opos;
		if (c > 0) memcpy(out + opos * oc, tmp, (size_t) c * oc * sizeof(sample_t));
		opos += c;
		pos += n;
	}
	for (;;) {
		const ssize_t f = refh_chain_drain(h, block_frames, tmp);
		if (f < 0) break;
		const ssize_t c = (opos + f <= out_cap) ? f : out_cap - opos;
		if (c > 0) memcpy(out + opos * oc, tmp, (size_t) c * oc * sizeof(sample_t));
		opos += c;
	}
	free(tmp);
	return opos;
}

/*
 * Two chains alive at once, interleaved on the SAME buffers -- the rebuild-with-crossfade of the CLI (dsp.c:1351-1366,
 * 1423-1430) and of `watch`: chain A runs alone for `switch_block` blocks; then chain B is built while A is still alive
 * and effects_chain_xfade_run (effects_chain.c:1241-1274) drives both on every block until the fade (xfade_frames) is
 * over; A is destroyed (finish_xfade, dsp.c:702-707), B runs on and is drained.  Both chains must keep rate and channels.
 * Returns output frames written, or -1.
 */
ssize_t refh_xfade_process(const char *chain_a, const char *chain_b, int fs, int channels, const char *dir,
                           const double *in, ssize_t frames, ssize_t block_frames, ssize_t switch_block, ssize_t xfade_frames,
                           double *out, ssize_t out_cap)
{
	struct effects_chain chain = EFFECTS_CHAIN_INITIALIZER;
	struct effects_chain_xfade_state xf = EFFECTS_CHAIN_XFADE_STATE_INITIALIZER;
	struct stream_info sa = { .fs = fs, .channels = channels }, sb = sa;
	if (build_effects_chain_from_string(chain_a, NULL, &chain, &sa, NULL, dir)) return -1;
	ssize_t len = get_effects_chain_buffer_len(&chain, block_frames, channels);
	sample_t *buf1 = calloc(len, sizeof(sample_t)), *buf2 = calloc(len, sizeof(sample_t));
	ssize_t pos = 0, opos = 0, blk = 0;
	const int oc = sa.channels;
	while (pos < frames) {
		if (blk == switch_block) {
			xf.chain[0] = chain;
			if (build_effects_chain_from_string(chain_b, NULL, &xf.chain[1], &sb, NULL, dir) || sb.fs != sa.fs || sb.channels != sa.channels) {
				destroy_effects_chain(&chain);
				if (sb.fs != sa.fs || sb.channels != sa.channels) destroy_effects_chain(&xf.chain[1]);
				free(buf1); free(buf2);
				return -1;
			}
			xf.frames = xf.pos = xfade_frames;
			const ssize_t len_b = get_effects_chain_buffer_len(&xf.chain[1], block_frames, channels);
			if (len_b > len) {
				len = len_b;
				free(buf1); free(buf2);
				buf1 = calloc(len, sizeof(sample_t)); buf2 = calloc(len, sizeof(sample_t));
			}
			xf.buf = calloc(len, sizeof(sample_t));
			if (xf.pos == 0) { destroy_effects_chain(&chain); chain = xf.chain[1]; effects_chain_xfade_reset(&xf); }
		}
		const ssize_t n = (frames - pos < block_frames) ? frames - pos : block_frames;
		memcpy(buf1, in + pos * channels, (size_t) n * channels * sizeof(sample_t));
		ssize_t w = n;
		sample_t *r;
		if (xf.pos > 0) {
			r = effects_chain_xfade_run(&xf, &w, buf1, buf2);
			if (xf.pos == 0) {
				destroy_effects_chain(&chain);
				chain = xf.chain[1];
				sample_t *b = xf.buf;
				effects_chain_xfade_reset(&xf);
				xf.buf = b;
			}
		}
		else r = run_effects_chain(&chain, &w, buf1, buf2);
		const ssize_t c = (opos + w <= out_cap) ? w : out_cap - opos;
		if (c > 0) memcpy(out + opos * oc, r, (size_t) c * oc * sizeof(sample_t));
		opos += (c > 0) ? c : 0;
		pos += n;
		++blk;
	}
	if (xf.pos > 0) {   /* input ended inside the fade: the new chain takes over, as finish_xfade() does */
		destroy_effects_chain(&chain);
		chain = xf.chain[1];
	}
	for (;;) {
		ssize_t f = block_frames;
		sample_t *r = drain_effects_chain(&chain, &f, buf1, buf2);
		if (f < 0) break;
		const ssize_t c = (opos + f <= out_cap) ? f : out_cap - opos;
		if (c > 0) memcpy(out + opos * oc, r, (size_t) c * oc * sizeof(sample_t));
		opos += (c > 0) ? c : 0;
	}
	destroy_effects_chain(&chain);
	free(buf1); free(buf2); free(xf.buf);
	return opos;
}

/* ---- CPU baseline (SURVEY.md section 8(d) "CPU baseline timing") ---- */

struct bench_arg {
	const char *chain_str, *dir;
	int fs, channels, first, count;
	ssize_t block_frames, n_blocks;
	const double *input;  /* one block, shared (read-only) */
	double elapsed;
	int err;
};

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static pthread_barrier_t bench_barrier;

static void *bench_worker(void *p)
{
	struct bench_arg *a = p;
	struct refh_chain **hs = calloc(a->count, sizeof(*hs));
	double *out = NULL;
	a->err = 0;
	for (int i = 0; i < a->count; ++i) {
		int ofs, och;
		hs[i] = refh_chain_new(a->chain_str, a->fs, a->channels, a->dir, &ofs, &och);
		if (!hs[i]) { a->err = 1; break; }
		if (!out) {
			const ssize_t mf = refh_chain_max_out_frames(hs[i], a->block_frames);
			out = malloc((size_t) (mf > a->block_frames ? mf : a->block_frames) * (och > a->channels ? och : a->channels) * sizeof(double));
		}
	}
	pthread_barrier_wait(&bench_barrier);
	const double t0 = now_s();
	if (!a->err) {
		for (ssize_t b = 0; b < a->n_blocks; ++b)
			for (int i = 0; i < a->count; ++i)
				if (refh_chain_run(hs[i], a->input, a->block_frames, out) < 0) a->err = 1;
	}
	a->elapsed = now_s() - t0;
	pthread_barrier_wait(&bench_barrier);
	for (int i = 0; i < a->count; ++i) refh_chain_free(hs[i]);
	free(hs);
	free(out);
	return NULL;
}

/*
 * Build n_streams independent chains, spread them over n_threads pthreads,
 * push n_blocks blocks of block_frames frames through every chain, time only
 * the run_effects_chain loops.  Returns wall seconds (max over threads), <0 on
 * error.  Input: one block (block_frames x channels doubles) shared by all.
 */
double refh_bench(const char *chain_str, const char *dir, int fs, int channels, int n_streams, int n_threads,
	ssize_t block_frames, ssize_t n_blocks, const double *input)
{
	if (n_threads < 1) n_threads = 1;
	if (n_threads > n_streams) n_threads = n_streams;
	pthread_t *th = calloc(n_threads, sizeof(*th));
	struct bench_arg *args = calloc(n_threads, sizeof(*args));
	pthread_barrier_init(&bench_barrier, NULL, n_threads);
	int first = 0;
	for (int t = 0; t < n_threads; ++t) {
		const int count = n_streams / n_threads + (t < n_streams % n_threads ? 1 : 0);
		args[t] = (struct bench_arg) { chain_str, dir, fs, channels, first, count, block_frames, n_blocks, input, 0.0, 0 };
		first += count;
		pthread_create(&th[t], NULL, bench_worker, &args[t]);
	}
	double worst = 0.0;
	int err = 0;
	for (int t = 0; t < n_threads; ++t) {
		pthread_join(th[t], NULL);
		if (args[t].elapsed > worst) worst = args[t].elapsed;
		err |= args[t].err;
	}
	pthread_barrier_destroy(&bench_barrier);
	free(th);
	free(args);
	return err ? -1.0 : worst;
}

int refh_ncpu(void)
{
	return (int) sysconf(_SC_NPROCESSORS_ONLN);
}
