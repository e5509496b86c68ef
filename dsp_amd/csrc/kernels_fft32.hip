// kernels_fft32.hip -- the FFT convolver's column and row kernels with a float32 working precision (W, H and the twiddle
// tables are float2): the arithmetic the reference itself uses behind `zita_convolver` (float32 in, float32 transforms,
// float32 out: zita_convolver.cpp:44,53,110).  Same source as the fp64 instance (fft_core.inc); half the bytes through W.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdint>
#include "kparams.h"
#include "fft_params.h"
#include "pcm_device.h"

namespace dspamd {
namespace p32 {
typedef float real;
#define FFT_F32 1
#include "fft_core.inc"
#undef FFT_F32
}  // namespace p32
}  // namespace dspamd
