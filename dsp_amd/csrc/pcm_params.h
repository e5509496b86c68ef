// pcm_params.h -- launch parameters of the wire-format kernels (kernels_pcm.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "kparams.h"

namespace dspamd {

struct PcmReadParams {
	const void *in;            // [S][in_stride_frames][C] samples of the wire format
	double *out;               // [S][out_stride_frames][C]
	long in_stride_frames, out_stride_frames, frames;
	int C, fmt;
};

struct PcmWriteParams {
	const double *in;          // [S][in_stride_frames][C]
	void *out;                 // [S][out_stride_frames][C] of the wire format
	long in_stride_frames, out_stride_frames, frames;
	int C, fmt;
	double dither_mult;        // 0 = no dither, else 1 / (PM_RAND_MAX 2^(prec-1))  (util.h:157-163)
	long samples_before;       // samples of each stream already written (position in the dither sequence)
	double *stats;             // optional [S][2]: clipped samples (as a 64-bit count), peak |sample|
};

void launch_pcm_read(const PcmReadParams &p, int n_streams, hipStream_t st);
void launch_pcm_write(const PcmWriteParams &p, int n_streams, hipStream_t st);

}  // namespace dspamd
