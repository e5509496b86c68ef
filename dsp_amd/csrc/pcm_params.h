// pcm_params.h -- launch parameters of the wire-format kernels (kernels_pcm.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace dspamd {

// same numbering as DSPAMD_PCM_* in include/dsp_amd.h
enum { PCM_U8 = 0, PCM_S8, PCM_S16, PCM_S24, PCM_S32, PCM_S24_3, PCM_FLOAT, PCM_DOUBLE, PCM_N_FORMATS };

struct PcmReadParams {
	const void *in;            // n samples of the wire format
	double *out;
	long n;
	int fmt;
};

struct PcmWriteParams {
	const double *in;          // [S][in_stride_frames][C]
	void *out;                 // packed [S][frames][C] of the wire format
	long in_stride_frames, frames;
	int C, fmt;
	double dither_mult;        // 0 = no dither, else 1 / (PM_RAND_MAX 2^(prec-1))  (util.h:157-163)
	long samples_before;       // samples of each stream already written (position in the dither sequence)
	double *stats;             // optional [S][2]: clipped samples (as a 64-bit count), peak |sample|
};

void launch_pcm_read(const PcmReadParams &p, hipStream_t st);
void launch_pcm_write(const PcmWriteParams &p, int n_streams, hipStream_t st);

}  // namespace dspamd
