"""CPU-only: the FIR that a time-reversed IIR effect (`biquad -r`, reverse_iir.c) is designed into by the product's host
code (dsp_amd/csrc/reverse_iir.cpp, through the C ABI's planning entry point -- no device involved), checked against the
real reference's comb-cascade implementation: the reference's output for any input must equal the convolution with that
FIR."""
import ctypes as C

import numpy as np
import pytest

from oracle_api import RefChain, rms

CHAINS = [
    ("lowpass -r 1k 0.707", 1),
    ("highpass -r 40 0.707", 1),
    ("lowpass_1 -r 2k", 1),
    ("highshelf -r60 8k 0.7 -3", 1),
    ("lowpass -r 1k 0.707 highpass -r 100 0.707", 1),      # merged into one effect
    ("lowpass -r 1k 0.5 lowpass -r 1k 0.5", 1),            # repeated poles: series states
    ("allpass -r 500 1.0 eq -r 2k 1.5 4", 1),
    ("biquad -r 0.2 0.3 0.1 1.0 -0.5 0.0", 1),             # one real pole, two zeros: FIR part of two taps
]


def plan_fir(chain, fs=48000, channels=1, effect=0, channel=0):
    import dsp_amd
    L = dsp_amd.load_library()
    delay = C.c_ssize_t(0)
    n = L.dspamd_plan_fir(chain.encode(), fs, channels, None, effect, channel, None, 0, C.byref(delay))
    assert n >= 0, L.dspamd_last_error()
    h = np.zeros(max(n, 1))
    n2 = L.dspamd_plan_fir(chain.encode(), fs, channels, None, effect, channel, h.ctypes.data, n, C.byref(delay))
    assert n2 == n
    return h[:n], int(delay.value)


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present (needs /root/reference at build time)")
@pytest.mark.parametrize("chain,n_effects", CHAINS)
def test_designed_fir_reproduces_the_reference(chain, n_effects):
    h, delay = plan_fir(chain)
    assert len(h) == delay + 1                 # 2^N + fir.n - 1 per state, summed (reverse_iir.c:623-625)
    r = RefChain(chain, 48000, 1)
    x = np.random.Generator(np.random.PCG64(7)).uniform(-0.5, 0.5, size=(3 * len(h) + 1000, 1))
    ref = r.process(x, block=4096)
    from scipy.signal import fftconvolve
    full = fftconvolve(x[:, 0], h)
    # the effect's own stream: the plain convolution (its delay is a REQUESTED negative delay, reverse_iir.c:275-280:
    # in a one-effect chain nothing else needs aligning), drained by `delay` extra frames (:234-239)
    assert len(ref) == len(x) + delay
    assert rms(ref[:, 0] - full[:len(ref)]) < 1e-13, rms(ref[:, 0] - full[:len(ref)])


def test_thresh_sets_the_length():
    h60, d60 = plan_fir("highpass -r20 40 0.707")
    h120, d120 = plan_fir("highpass -r200 40 0.707")
    assert d120 > d60 and len(h120) == d120 + 1


def test_bad_thresh_is_refused():
    import dsp_amd
    L = dsp_amd.load_library()
    assert L.dspamd_plan_fir(b"lowpass -r5 1k 0.707", 48000, 1, None, 0, 0, None, 0, None) == -1
