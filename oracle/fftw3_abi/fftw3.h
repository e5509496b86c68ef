/*
 * oracle/fftw3_abi/fftw3.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Declarations of the 11-function subset of the FFTW3 double-precision API
 * that the reference calls (fir.c:127-132,329-352; fir_p.c:72-87,469-492;
 * resample.c:113-133,336-366; util.c:484,495).  FFTW3 itself is a
 * third-party dependency that is NOT vendored in /root/reference and is NOT
 * installed in this image (no version is pinned by the reference: configure:137
 * only tests `pkg-config --exists fftw3`).  This header restates FFTW's
 * published C ABI so the reference sources compile unmodified; the
 * implementation behind it is oracle/fftw3_abi/minifftw.c (own mixed-radix
 * fp64 FFT) or, optionally, Intel MKL's FFTW3 wrapper.
 */
#ifndef ORACLE_FFTW3_ABI_H
#define ORACLE_FFTW3_ABI_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* FFTW: if <complex.h> was included first, fftw_complex is the native type */
#if !defined(__cplusplus) && defined(_Complex_I) && defined(complex) && defined(I)
typedef double _Complex fftw_complex;
#else
typedef double fftw_complex[2];
#endif

typedef struct fftw_plan_s *fftw_plan;

#define FFTW_MEASURE  (0U)
#define FFTW_ESTIMATE (1U << 6)

void *fftw_malloc(size_t n);
void fftw_free(void *p);
fftw_plan fftw_plan_dft_r2c_1d(int n, double *in, fftw_complex *out, unsigned flags);
fftw_plan fftw_plan_dft_c2r_1d(int n, fftw_complex *in, double *out, unsigned flags);
void fftw_execute(const fftw_plan p);
void fftw_execute_dft_r2c(const fftw_plan p, double *in, fftw_complex *out);
void fftw_execute_dft_c2r(const fftw_plan p, fftw_complex *in, double *out);
void fftw_destroy_plan(fftw_plan p);
int fftw_import_wisdom_from_filename(const char *filename);
int fftw_export_wisdom_to_filename(const char *filename);

#ifdef __cplusplus
}
#endif
#endif
