// resample_params.h -- launch parameters of the polyphase resampler
#pragma once
#include <hip/hip_runtime.h>
#include "kparams.h"

namespace dspamd {

struct ResampleParams {
	const double *ring;        // [S][ring_len][C] input history (interleaved frames)
	long ring_len, ring_mask;
	long q_total;              // input frames received so far (frames >= q_total read as zero: drain)
	const double *tab;         // [J][n] polyphase taps, gain folded in
	int n, d, J, C, KT;
	long out_delay;            // leading full-rate outputs dropped (resample.c:144-147)
	long m_first, m_count;     // visible output frames [m_first, m_first + m_count) to produce
	double *out;               // [S][out_stride][C]
	long out_stride_frames, out_frame0;
	WireSink sink;             // last stage of a pipeline run in wire formats: `out` holds samples of sink.fmt (sink_bs bytes each)
	int sink_bs;
};

// rational n/d as a GEMM on the fp64 matrix cores (kernels_resample.hip, resample_gemm_kernel)
struct ResampleGemmParams {
	const double *ring;        // [S][ring_len][C]
	long ring_len, ring_mask, q_total;
	const double *G;           // [Kpad][Npad] (zero padded): G[u + J - 1][r] = tab[floor(r d / n) - u][(r d) mod n]
	int NB, DB, J, Kpad, Npad; // outputs / inputs per block, taps, padded K and N
	int C, log2cp;             // channels, log2 of the channel count padded to a power of two (<= 3)
	long out_delay, m_first, m_count, i_first;   // visible frames [m_first, m_first + m_count); first block index
	double *out;
	long out_stride_frames, out_frame0;
	WireSink sink;             // as in ResampleParams
	int sink_bs;
};

size_t resample_gemm_lds_bytes(int DB, int J, int log2cp);
void launch_resample_gemm(const ResampleGemmParams &p, int n_streams, hipStream_t st);
void launch_resample(const ResampleParams &p, int n_streams, hipStream_t st);
void launch_resample_push(const double *in, long in_stride, double *ring, long ring_len, long pos, long frames, int C, int n_streams, hipStream_t st);

}  // namespace dspamd
