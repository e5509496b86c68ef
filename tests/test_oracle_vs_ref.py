"""Pin the CPU restatement (oracle/dsp_oracle.c) against the REAL reference
(oracle/_ref/libdspref.so, compiled by oracle/Makefile from /root/reference).

CPU-only.  Skipped when the reference build is absent (it is git-ignored; it is
rebuilt by __graft_entry__.build() whenever /root/reference exists).
"""
import os

import numpy as np
import pytest

from oracle_api import Oracle, RefChain, rms

pytestmark = pytest.mark.skipif(not (Oracle.available() and RefChain.available()), reason="oracle/_ref or liboracle.so not built")

FS = 48000
O = Oracle

# (chain text, oracle design args)  -- biquad.h:30-60, biquad.c:441-562
BIQUADS = [
    ("lowpass_1 1k", (O and 1, 1000.0, 0, 0, 0, 1)),
    ("highpass_1 300", (2, 300.0, 0, 0, 0, 1)),
    ("allpass_1 2k", (3, 2000.0, 0, 0, 0, 1)),
    ("lowshelf_1 200 4", (4, 200.0, 0, 4.0, 0, 1)),
    ("highshelf_1 5k -3", (5, 5000.0, 0, -3.0, 0, 1)),
    ("lowpass_1p 800", (6, 800.0, 0, 0, 0, 1)),
    ("lowpass 1k 0.707", (7, 1000.0, 0.707, 0, 0, 1)),
    ("highpass 20 0.707", (8, 20.0, 0.707, 0, 0, 1)),
    ("bandpass_skirt 1k 2", (9, 1000.0, 2.0, 0, 0, 1)),
    ("bandpass_peak 1k 1o", (10, 1000.0, 1.0, 0, 0, 4)),
    ("notch 60 10", (11, 60.0, 10.0, 0, 0, 1)),
    ("allpass 500 200h", (12, 500.0, 200.0, 0, 0, 5)),
    ("eq 100 1.0 3", (13, 100.0, 1.0, 3.0, 0, 1)),
    ("eq 3200 1k -2.5", (13, 3200.0, 1000.0, -2.5, 0, 5)),
    ("lowshelf 100 0.7 6", (14, 100.0, 0.7, 6.0, 0, 1)),
    ("lowshelf 100 0.8s 6", (14, 100.0, 0.8, 6.0, 0, 2)),
    ("highshelf 8k 0.7 -3", (15, 8000.0, 0.7, -3.0, 0, 1)),
    ("highshelf 8k 6d -3", (15, 8000.0, 6.0, -3.0, 0, 3)),
    ("lowpass_transform 80 0.9 40 0.5", (16, 80.0, 0.9, 40.0, 0.5, 1)),
    ("linkwitz_transform 80 0.9 40 0.5", (17, 80.0, 0.9, 40.0, 0.5, 1)),
]


def noise(frames, ch, seed=1234, amp=0.5):
    return np.random.Generator(np.random.PCG64(seed)).uniform(-amp, amp, size=(frames, ch))


@pytest.mark.parametrize("chain,args", BIQUADS)
def test_biquad_types_bitexact(chain, args):
    x = noise(3000, 2)
    ref = RefChain(chain, FS, 2).process(x, block=777)
    t, a0, a1, a2, a3, wt = args
    c = O.biquad_design(t, FS, a0, a1, a2, a3, wt)
    y, _ = O.biquad_run(c, x.copy())
    assert ref.shape == y.shape
    assert np.array_equal(ref, y), f"max diff {np.abs(ref - y).max()}"


def test_biquad_bw_macro_and_raw():
    w, t, ok = O.parse_width("bw4.1")
    assert ok and t == 1
    x = noise(1000, 1)
    ref = RefChain("lowpass 2k bw4.1", FS, 1).process(x)
    y, _ = O.biquad_run(O.biquad_design(7, FS, 2000.0, w, 0, 0, 1), x.copy())
    assert np.array_equal(ref, y)
    ref = RefChain("biquad 0.2 0.3 0.1 1.1 -0.4 0.2", FS, 1).process(x)
    y, _ = O.biquad_run(O.biquad_coefs(0.2, 0.3, 0.1, 1.1, -0.4, 0.2), x.copy())
    assert np.array_equal(ref, y)
    # deemph preset (biquad.c:503-521)
    ref = RefChain("deemph", FS, 1).process(x)
    y, _ = O.biquad_run(O.biquad_design(15, FS, 5356.0, 0.479, -9.62, 0, 2), x.copy())
    assert np.array_equal(ref, y)


def test_gain_add_remix_delay_bitexact():
    x = noise(500, 4)
    ref = RefChain("gain -6 :1,3 mult 0.3 : add 0.001", FS, 4).process(x)
    y = x.copy()
    g = np.full(4, 10 ** (-6 / 20.0)) * np.array([1, 0.3, 1, 0.3])
    O.lib().orc_gain_run(y.ctypes.data, 500, 4, g.ctypes.data)
    a = np.full(4, 0.001)
    O.lib().orc_add_run(y.ctypes.data, 500, 4, a.ctypes.data)
    assert np.array_equal(ref, y)

    ref = RefChain("remix 0,1 2 . 1,2,3", FS, 4).process(x)
    sel = np.zeros((4, 4), dtype=np.int8)
    sel[0, [0, 1]] = 1; sel[1, 2] = 1; sel[3, [1, 2, 3]] = 1
    out = np.zeros((500, 4))
    O.lib().orc_remix_run(x.ctypes.data, out.ctypes.data, 500, 4, 4, sel.ctypes.data)
    assert np.array_equal(ref, out)

    rc = RefChain(":1 delay 37S", FS, 2)
    ref = rc.process(x[:, :2].copy(), block=100)
    y = np.ascontiguousarray(x[:, :2])
    y = np.vstack([y, np.zeros((rc.drain_frames() if False else 37, 2))])
    ring = np.zeros(37); p = __import__("ctypes").c_ssize_t(0)
    O.lib().orc_delay_run(y.ctypes.data + 8, len(y), 2, ring.ctypes.data, 37, __import__("ctypes").byref(p))
    assert ref.shape == y.shape and np.array_equal(ref, y)


def _write_filter(tmp_path, taps, name="f.raw"):
    p = os.path.join(str(tmp_path), name)
    np.asarray(taps, dtype="<f8").tofile(p)
    return p


def make_filter(n, seed=7, decay=None):
    rng = np.random.default_rng(seed)
    h = rng.standard_normal(n) * np.exp(-np.arange(n) / (decay or max(n / 8.0, 1.0)))
    return h / np.sqrt(np.sum(h * h)) / 4.0


def test_fir_direct_bitexact():
    x = noise(400, 2)
    taps = make_filter(11)
    txt = "coefs:" + ",".join(repr(float(t)) for t in taps)
    ref = RefChain(f"fir {txt}", FS, 2).process(x, block=64)
    y = O.per_channel("fir_direct", taps, np.vstack([x, np.zeros((10, 2))]))
    assert ref.shape == y.shape and np.array_equal(ref, y)


@pytest.mark.parametrize("ntaps", [33, 100, 1000, 4095, 4096, 20000])
def test_fir_p_matches_ref(tmp_path, ntaps):
    x = noise(9000, 2, seed=5)
    taps = make_filter(ntaps)
    path = _write_filter(tmp_path, taps)
    ref = RefChain(f"fir_p -t pcm -e double -c 1 {path}", FS, 2).process(x, block=1000)
    xin = np.vstack([x, np.zeros((ntaps - 1, 2))])
    y = O.per_channel("fir_p", taps, xin, 0)
    assert ref.shape == y.shape
    # same algorithm, same FFT: bit-identical except for thread scheduling (none here)
    assert rms(ref - y) <= 1e-17 * 10, rms(ref - y)
    y2 = np.stack([O.conv_full(x[:, k], taps) for k in range(2)], axis=1)
    assert rms(ref - y2) < 1e-15


@pytest.mark.parametrize("ntaps", [17, 1000])
def test_fir_matches_ref(tmp_path, ntaps):
    x = noise(5000, 1, seed=9)
    taps = make_filter(ntaps)
    path = _write_filter(tmp_path, taps)
    ref = RefChain(f"fir -t pcm -e double -c 1 {path}", FS, 1).process(x, block=512)
    # CLI semantics: the len-frame latency is discarded by the end-of-chain align (align.c:147-152)
    L = O.lib().orc_next_fast_fftw_len(ntaps)
    xin = np.vstack([x, np.zeros((L + ntaps - 1, 1))])
    y = O.per_channel("fir", taps, xin)[L:]
    assert ref.shape == y.shape
    assert rms(ref - y) < 1e-16


def test_fir_p_plan_table():
    # SURVEY.md appendix A.1 (taken from the reference's -v log)
    import ctypes as C
    def plan(n, single):
        l = (C.c_int * 4)(); k = (C.c_int * 4)(); d = (C.c_int * 4)()
        ng = O.lib().orc_fir_p_plan(n, 0, single, l, k, d)
        return [(l[i], k[i], d[i]) for i in range(ng)]
    assert plan(1000, 1) == [(32, 3, 0), (128, 7, 0)]
    assert plan(4095, 1) == [(32, 3, 0), (128, 3, 0), (512, 7, 0)]
    assert plan(4096, 0) == [(32, 7, 0), (128, 6, 128), (512, 6, 512)]
    assert plan(65536, 0) == [(32, 15, 0), (256, 14, 256), (2048, 30, 2048)]
    assert plan(131072, 0) == [(32, 15, 0), (256, 30, 256), (4096, 30, 4096)]


@pytest.mark.parametrize("fs_in,fs_out", [(48000, 96000), (96000, 48000), (44100, 48000), (48000, 44100)])
def test_resample_matches_ref(fs_in, fs_out):
    x = noise(6000, 2, seed=3, amp=0.4)
    ref = RefChain(f"resample {fs_out}", fs_in, 2).process(x, block=1000)
    y = O.resample(x, fs_in, fs_out, block=1000)
    assert ref.shape == y.shape, (ref.shape, y.shape)
    assert rms(ref - y) < 1e-16
    assert ref.shape[0] == -(-6000 * fs_out // fs_in)


def test_resample_params_table():
    # SURVEY.md appendix A.2
    import ctypes as C
    def params(a, b):
        st = O.lib().orc_resample_new(a, b, 0.939)
        p = np.zeros(8, dtype=np.int32)
        O.lib().orc_resample_params(st, p.ctypes.data)
        O.lib().orc_resample_free(st)
        return [int(v) for v in p[:6]]
    assert params(48000, 96000) == [2, 1, 1166, 588, 1176, 583]
    assert params(96000, 48000) == [1, 2, 1166, 1176, 588, 292]
    assert params(44100, 48000) == [160, 147, 635, 588, 640, 317]


def test_hilbert_matches_ref():
    x = noise(3000, 1, seed=11)
    taps = O.hilbert_taps(255)
    ref = RefChain("hilbert -p 255", FS, 1).process(x, block=500)
    y = O.per_channel("fir_p", taps, np.vstack([x, np.zeros((254, 1))]), 0)
    assert ref.shape == y.shape and rms(ref - y) < 1e-16


def test_sgen_sine_matches_cli(tmp_path):
    import subprocess
    cli = os.path.join(os.path.dirname(RefChain.so_path()), "dsp_ref")
    out = os.path.join(str(tmp_path), "s.raw")
    subprocess.run([cli, "-q", "-t", "sgen", "-r", "48k", "-c", "2", "sine:freq=1234.5+4800S", "-o", "-t", "pcm", "-e", "double", out],
                   check=True, stderr=subprocess.DEVNULL)
    ref = np.fromfile(out).reshape(-1, 2)
    y = O.sgen_sine(4800, 2, 48000, 1234.5)
    assert ref.shape == y.shape and np.array_equal(ref, y)


@pytest.mark.parametrize("chain,ch", [("st2ms :1 gain -2 : ms2st", 2), (":0,2 st2ms : lowpass 1k 0.7 :0,2 ms2st", 3),
                                      ("crossfeed 500 6", 2), (":1,2 crossfeed 1.2k 3 : eq 300 1 2", 4)])
def test_pair_effects_bitexact(chain, ch):
    # the restatement of st2ms / ms2st (st2ms.c:28-54) and crossfeed (crossfeed.c:33-50) against the real reference, bit for bit
    import oracle_chain
    x = noise(2000, ch)
    ref = RefChain(chain, FS, ch).process(x, block=512)
    y, _ = oracle_chain.run(chain, x, FS)
    assert y.shape == ref.shape and np.array_equal(ref, y)


@pytest.mark.parametrize("chain,ch", [("delay 37S", 2), ("delay 10S :0 delay 72S", 2), ("delay -f 0.37S", 2), (":0 delay -f 0.37S :1 delay -f5 7.3S : eq 500 1.0 2", 2),
                                      ("delay -f1 0.5S :1 delay 3S", 2), ("delay 0.2m :0 delay -f 1.5m", 2), ("delay -f 0.37S :1 delay 5S", 2),
                                      (":0,2 delay -f9 11.7S :1 delay -f 0.05S : lowpass 3k 0.7", 3), ("gain -3 :2 delay 1m : delay -f16 20.5S", 3)])
def test_delays_bitexact(chain, ch):
    # integer and fractional delays (delay.c:126-205, allpass.h:46-118; realised through the host's alignment, align.c:125-146):
    # merged amounts, default and explicit all-pass orders, requests that go negative -- bit for bit, lengths included
    import oracle_chain
    x = noise(1500, ch)
    ref = RefChain(chain, FS, ch).process(x, block=500)
    y, _ = oracle_chain.run(chain, x, FS)
    assert y.shape == ref.shape and np.array_equal(ref, y)


def test_sgen_sweep_and_delta_vs_reference_cli(tmp_path):
    """sgen.c:46-67, 163: the oracle's restatement of the swept sine and of the impulse against the stock CLI's own generator
    (same libm: bit for bit)."""
    import os
    import subprocess
    from oracle_api import REF_DIR
    exe = os.path.join(REF_DIR, "dsp_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/dsp_ref not built")
    o = os.path.join(str(tmp_path), "o.raw")
    for spec, ref in (("sine:freq=100-8000+1", lambda: Oracle.sgen_sweep(48000, 2, 48000, 100.0, 8000.0, 48000)),
                      ("sine:freq=3k-50+0.5", lambda: Oracle.sgen_sweep(24000, 2, 48000, 3000.0, 50.0, 24000)),
                      ("delta:offset=100S+0.25", lambda: Oracle.sgen_delta(12000, 2, 100))):
        r = subprocess.run([exe, "-q", "-t", "sgen", "-r", "48k", "-c", "2", spec, "-o", "-t", "pcm", "-e", "double", o], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
        assert r.returncode == 0, r.stderr[-500:]
        y = np.fromfile(o).reshape(-1, 2)
        want = ref()
        assert y.shape == want.shape, (spec, y.shape, want.shape)
        assert np.array_equal(y, want), spec
