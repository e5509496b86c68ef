/*
 * oracle/fftw3_abi/minifftw.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A small self-written fp64 mixed-radix FFT behind the FFTW3 ABI subset declared
 * in fftw3.h, so that the *unmodified* reference sources (fir.c, fir_p.c,
 * resample.c, util.c) can be compiled into oracle/_ref without the real FFTW3
 * (absent from this image, un-vendored and un-pinned by the reference).
 *
 * Semantics restated from FFTW's documentation:
 *   r2c: X[k] = sum_j x[j] exp(-2 pi i j k / n), k = 0..n/2  (unnormalised)
 *   c2r: x[j] = sum_k X[k] exp(+2 pi i j k / n) over the Hermitian-extended
 *        spectrum (unnormalised; c2r(r2c(x)) = n x); the input array may be
 *        destroyed (FFTW's default for out-of-place c2r).
 * Only even n is supported (every call site in the reference uses 2*len).
 *
 * Algorithm: real transform of size n through a complex transform of size n/2
 * plus the usual split step; the complex transform is a recursive
 * decimation-in-time mixed-radix FFT (radix 4/2/3/5/7, generic O(p^2) butterfly
 * for any other prime).  Execution is thread-safe (scratch is thread-local).
 */
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fftw3.h"

typedef double _Complex cplx;

#define MAX_FACTORS 40

struct cfft {
	int n, nf;
	int radix[MAX_FACTORS], rem[MAX_FACTORS];
	cplx *tw;  /* tw[k] = exp(-2 pi i k / n), k < n */
};

struct fftw_plan_s {
	int n, h, kind;  /* kind 0: r2c, 1: c2r */
	double *rbuf;
	cplx *cbuf;
	struct cfft cf;  /* size h = n/2 */
	cplx *wr;        /* wr[k] = exp(-2 pi i k / n), k <= h */
};

static __thread cplx *tls_scratch = NULL;
static __thread size_t tls_scratch_len = 0;

static cplx *get_scratch(size_t len)
{
	if (len > tls_scratch_len) {
		free(tls_scratch);
		tls_scratch = NULL;
		if (posix_memalign((void **) &tls_scratch, 64, len * sizeof(cplx)) != 0) {
			fprintf(stderr, "minifftw: out of memory\n");
			abort();
		}
		tls_scratch_len = len;
	}
	return tls_scratch;
}

static void cfft_init(struct cfft *c, int n)
{
	c->n = n;
	c->nf = 0;
	int m = n;
	while (m > 1) {
		int p;
		if (m % 4 == 0) p = 4;
		else if (m % 2 == 0) p = 2;
		else if (m % 3 == 0) p = 3;
		else if (m % 5 == 0) p = 5;
		else if (m % 7 == 0) p = 7;
		else {
			p = 11;
			while (m % p != 0) {
				p += 2;
				if ((long) p * p > m) { p = m; break; }
			}
		}
		m /= p;
		c->radix[c->nf] = p;
		c->rem[c->nf] = m;
		++c->nf;
	}
	if (c->nf == 0) {  /* n == 1 */
		c->radix[0] = 1;
		c->rem[0] = 1;
		c->nf = 1;
	}
	c->tw = malloc((size_t) n * sizeof(cplx));
	for (int k = 0; k < n; ++k) {
		const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double) k / (long double) n;
		c->tw[k] = (double) cosl(a) + I * (double) sinl(a);
	}
}

static inline cplx twiddle(const struct cfft *c, long idx, int inverse)
{
	const cplx w = c->tw[idx % c->n];
	return inverse ? conj(w) : w;
}

static void butterfly(const struct cfft *c, cplx *out, int fstride, int m, int p, int inverse)
{
	if (p == 2) {
		for (int k = 0; k < m; ++k) {
			const cplx t = out[m + k] * twiddle(c, (long) k * fstride, inverse);
			out[m + k] = out[k] - t;
			out[k] += t;
		}
	}
	else if (p == 4) {
		for (int k = 0; k < m; ++k) {
			const cplx a0 = out[k];
			const cplx a1 = out[m + k] * twiddle(c, (long) k * fstride, inverse);
			const cplx a2 = out[2*m + k] * twiddle(c, 2L * k * fstride, inverse);
			const cplx a3 = out[3*m + k] * twiddle(c, 3L * k * fstride, inverse);
			const cplx s02 = a0 + a2, d02 = a0 - a2;
			const cplx s13 = a1 + a3, d13 = a1 - a3;
			/* forward: multiply d13 by -i; inverse: by +i */
			const cplx jd = inverse ? (-cimag(d13) + I * creal(d13)) : (cimag(d13) - I * creal(d13));
			out[k] = s02 + s13;
			out[m + k] = d02 + jd;
			out[2*m + k] = s02 - s13;
			out[3*m + k] = d02 - jd;
		}
	}
	else {
		cplx t[64], *tp = t;
		if (p > 64) tp = malloc((size_t) p * sizeof(cplx));
		const int pstride = c->n / p;  /* w_p = tw[pstride] */
		for (int k = 0; k < m; ++k) {
			for (int q = 0; q < p; ++q)
				tp[q] = out[q*m + k] * twiddle(c, (long) q * k * fstride, inverse);
			for (int r = 0; r < p; ++r) {
				cplx acc = tp[0];
				for (int q = 1; q < p; ++q)
					acc += tp[q] * twiddle(c, (long) ((q * r) % p) * pstride, inverse);
				out[r*m + k] = acc;
			}
		}
		if (tp != t) free(tp);
	}
}

static void cfft_work(const struct cfft *c, cplx *out, const cplx *in, int fstride, int level, int inverse)
{
	const int p = c->radix[level], m = c->rem[level];
	if (p == 1) {
		out[0] = in[0];
		return;
	}
	if (m == 1) {
		for (int q = 0; q < p; ++q)
			out[q] = in[(long) q * fstride];
	}
	else {
		for (int q = 0; q < p; ++q)
			cfft_work(c, out + (long) q * m, in + (long) q * fstride, fstride * p, level + 1, inverse);
	}
	butterfly(c, out, fstride, m, p, inverse);
}

void *fftw_malloc(size_t n)
{
	void *p = NULL;
	if (posix_memalign(&p, 64, n ? n : 1) != 0) return NULL;
	return p;
}

void fftw_free(void *p)
{
	free(p);
}

static fftw_plan plan_new(int n, int kind, double *r, cplx *c)
{
	if (n < 2 || (n & 1)) {
		fprintf(stderr, "minifftw: only even n >= 2 supported (n=%d)\n", n);
		return NULL;
	}
	fftw_plan p = calloc(1, sizeof(*p));
	if (!p) return NULL;
	p->n = n;
	p->h = n / 2;
	p->kind = kind;
	p->rbuf = r;
	p->cbuf = c;
	cfft_init(&p->cf, p->h);
	p->wr = malloc((size_t) (p->h + 1) * sizeof(cplx));
	for (int k = 0; k <= p->h; ++k) {
		const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double) k / (long double) n;
		p->wr[k] = (double) cosl(a) + I * (double) sinl(a);
	}
	return p;
}

fftw_plan fftw_plan_dft_r2c_1d(int n, double *in, fftw_complex *out, unsigned flags)
{
	(void) flags;
	return plan_new(n, 0, in, (cplx *) out);
}

fftw_plan fftw_plan_dft_c2r_1d(int n, fftw_complex *in, double *out, unsigned flags)
{
	(void) flags;
	return plan_new(n, 1, out, (cplx *) in);
}

void fftw_execute_dft_r2c(const fftw_plan p, double *in, fftw_complex *out_)
{
	cplx *out = (cplx *) out_;
	const int h = p->h;
	cplx *z = get_scratch((size_t) h);
	/* the real array viewed as h complex numbers is exactly z[j] = x[2j] + i x[2j+1] */
	cfft_work(&p->cf, z, (const cplx *) in, 1, 0, 0);
	const cplx z0 = z[0];
	out[0] = creal(z0) + cimag(z0);
	out[h] = creal(z0) - cimag(z0);
	for (int k = 1; k < h; ++k) {
		const cplx a = z[k], b = conj(z[h - k]);
		const cplx e = 0.5 * (a + b);
		const cplx o = 0.5 * (a - b);  /* = i O[k] */
		/* X[k] = E[k] + w^k O[k] = e - i w^k o */
		const cplx wo = p->wr[k] * o;
		out[k] = e + (cimag(wo) - I * creal(wo));
	}
}

void fftw_execute_dft_c2r(const fftw_plan p, fftw_complex *in_, double *out)
{
	const cplx *in = (const cplx *) in_;
	const int h = p->h;
	cplx *z = get_scratch((size_t) h * 2);
	cplx *zin = z + h;
	for (int k = 0; k < h; ++k) {
		const cplx a = (k == 0) ? creal(in[0]) : in[k];
		const cplx b = (k == 0) ? creal(in[h]) : conj(in[h - k]);
		const cplx s = a + b, d = a - b;
		const cplx wd = conj(p->wr[k]) * d;
		/* Z[k] = s + i conj(w^k) d */
		zin[k] = s + (-cimag(wd) + I * creal(wd));
	}
	cfft_work(&p->cf, z, zin, 1, 0, 1);
	memcpy(out, z, (size_t) h * sizeof(cplx));
}

void fftw_execute(const fftw_plan p)
{
	if (p->kind == 0) fftw_execute_dft_r2c(p, p->rbuf, (fftw_complex *) p->cbuf);
	else fftw_execute_dft_c2r(p, (fftw_complex *) p->cbuf, p->rbuf);
}

void fftw_destroy_plan(fftw_plan p)
{
	if (!p) return;
	free(p->cf.tw);
	free(p->wr);
	free(p);
}

int fftw_import_wisdom_from_filename(const char *filename)
{
	(void) filename;
	return 0;
}

int fftw_export_wisdom_to_filename(const char *filename)
{
	(void) filename;
	return 0;
}
