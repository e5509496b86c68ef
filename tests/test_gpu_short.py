"""The one-trip convolver for short filters behind long calls (kernels_short.hip, round 5): filters of up to 4097 taps on calls of at least 1024 frames
(up to 8193 taps where the calls fill the larger window's blocks) --
a pair's 8192- or 16384-point transform in one workgroup's LDS (the larger window from about 2300 taps on where the calls fill its blocks; either one
forced with DSP_AMD_CONV_SHORT=13 / 14), one read of the window and one write of the outputs per block -- against the real reference and
against the four-step transforms (DSP_AMD_CONV_SHORT=0) on the same inputs: tap counts at both ends, ragged call sequences, drains, per-channel filters,
selectors, `fir`'s latency, a stage that feeds the next convolver's rings (BASELINE config 5's shape in small), a cascade in front, reset."""
import os

import numpy as np
import pytest

from oracle_api import RefChain, rms

pytestmark = pytest.mark.gpu


def make_filter(n, seed=7, decay=600.0):
    rng = np.random.default_rng(seed)
    h = rng.standard_normal(n) * np.exp(-np.arange(n) / decay)
    return h / np.sqrt(np.sum(h * h)) / 4.0


@pytest.fixture(scope="module")
def amd():
    import dsp_amd
    assert dsp_amd.load_library().dspamd_device_count() >= 1
    return dsp_amd


def build(amd, chain, C, S, B, short):
    """short: False = the four-step transforms, True = the planner's own window, 13 / 14 = that window"""
    if short is not True:
        os.environ["DSP_AMD_CONV_SHORT"] = str(int(short))
    try:
        return amd.BatchChain(chain, 48000, C, S, B)
    finally:
        os.environ.pop("DSP_AMD_CONV_SHORT", None)


def run_calls(b, x, sizes):
    import torch
    outs, pos = [], 0
    for n in sizes:
        outs.append(b.run(x[:, pos:pos + n, :].contiguous()).clone())
        pos += n
    while True:
        o = b.drain(max(sizes))
        if o is None:
            break
        outs.append(o.clone())
    return torch.cat([o for o in outs if o.shape[1]], dim=1)


CASES = [
    # (chain with {F}, taps, filter channels, S, C, call sizes, window: True = the planner's choice (its size stated), 13 / 14 = forced)
    ("fir_p -t pcm -e double -c 1 {F}", 4095, 1, 9, 2, (20000, 20000, 4097, 1024, 30001), (True, 8192)),
    ("fir_p -t pcm -e double -c 1 {F}", 4095, 1, 9, 2, (20000, 20000, 4097, 1024, 30001), (14, 16384)),   # ragged calls on the larger window
    ("fir_p -t pcm -e double -c 1 {F}", 4097, 1, 64, 8, (65536, 65536, 1), (True, 16384)),
    ("fir_p -t pcm -e double -c 1 {F}", 4097, 1, 64, 8, (65536, 65536, 1), (13, 8192)),
    ("fir_p -t pcm -e double -c 1 {F}", 33, 1, 5, 4, (8192, 5000, 8192), (True, 8192)),
    ("fir_p -t pcm -e double -c 1 {F}", 33, 1, 5, 4, (8192, 5000, 8192), (14, 16384)),
    ("fir_p -t pcm -e double -c 1 {F}", 1000, 1, 3, 3, (12288, 12288, 777), (True, 8192)),                      # an odd channel count: a pair with one channel
    ("fir_p -t pcm -e double -c 1 {F}", 2400, 1, 3, 3, (60000, 12288, 777), (True, 16384)),
    ("fir_p -t pcm -e double -c 1 {F}", 2200, 1, 3, 3, (60000, 12288, 777), (True, 8192)),
    ("fir_p -t pcm -e double -c 4 {F}", 2500, 4, 6, 4, (16384, 9999, 16384), (True, 8192)),                     # one filter per channel
    ("fir_p -t pcm -e double -c 4 {F}", 2500, 4, 6, 4, (16384, 9999, 16384), (14, 16384)),
    (":0,2 fir_p -t pcm -e double -c 1 {F}", 3000, 1, 4, 4, (16384, 16384), (True, 8192)),                      # two of four channels convolved, the others passed through
    (":0,2 fir_p -t pcm -e double -c 1 {F}", 3000, 1, 4, 4, (57344, 16384), (True, 16384)),
    ("fir -t pcm -e double -c 1 {F}", 2000, 1, 4, 2, (10000, 10000, 3000), (True, 8192)),                        # `fir`: the same values, a transform length late
    ("fir -t pcm -e double -c 1 {F}", 2000, 1, 4, 2, (10000, 10000, 3000), (14, 16384)),
    ("lowpass 1k 0.707 eq 400 2.0 1.5 fir_p -t pcm -e double -c 1 {F} gain -2", 4000, 1, 16, 8, (32768, 1500, 32768), (True, 8192)),   # a cascade writes the rings
    ("lowpass 1k 0.707 eq 400 2.0 1.5 fir_p -t pcm -e double -c 1 {F} gain -2", 4000, 1, 16, 8, (65536, 1500, 32768), (True, 16384)),
    # 4098 ... 8193 taps: the larger window only, and only where the calls fill its blocks
    ("fir_p -t pcm -e double -c 1 {F}", 8193, 1, 6, 2, (65536, 40000, 65536, 3), (True, 16384)),
    ("fir -t pcm -e double -c 1 {F}", 6000, 1, 4, 4, (49152, 1024, 20000), (True, 16384)),
    ("fir_p -t pcm -e double -c 4 {F}", 5000, 4, 5, 4, (60000, 60000), (True, 16384)),
]


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
@pytest.mark.parametrize("chain,taps,fch,S,C,sizes,window", CASES)
def test_one_trip_vs_reference_and_four_step(amd, tmp_path, chain, taps, fch, S, C, sizes, window):
    import torch
    f = os.path.join(str(tmp_path), "h.raw")
    h = np.stack([make_filter(taps, seed=taps + c) for c in range(fch)], axis=1)
    np.asarray(h, dtype="<f8").tofile(f)
    chain = chain.replace("{F}", f)
    bo, bs = build(amd, chain, C, S, max(sizes), window[0]), build(amd, chain, C, S, max(sizes), False)
    assert "one-trip" in bo.plan() and f"N={window[1]}=" in bo.plan(), bo.plan()
    assert "one-trip" not in bs.plan(), bs.plan()
    g = torch.Generator(device="cuda"); g.manual_seed(taps)
    x = torch.rand((S, sum(sizes), C), dtype=torch.float64, device="cuda", generator=g) - 0.5
    L = amd.load_library()
    L.dspamd_profile_enable(1)
    yo = run_calls(bo, x, sizes)
    names = {ln.split()[0] for ln in L.dspamd_profile_collect().decode().splitlines()}
    L.dspamd_profile_enable(0)
    assert "conv_short" in names and not (names & {"conv_col_fwd", "conv_row", "conv_col_inv"}), names
    ys = run_calls(bs, x, sizes)
    assert yo.shape == ys.shape
    assert float((yo - ys).pow(2).mean().sqrt()) < 1e-13
    for s in sorted({0, S - 1, S // 2}):
        ref = RefChain(chain, 48000, C).process(x[s].cpu().numpy(), block=4096)
        got = yo[s].cpu().numpy()
        assert ref.shape == got.shape, (ref.shape, got.shape)
        assert rms(ref - got) < 1e-12, (s, rms(ref - got))
    # reset: the stream starts over, same bits; every stream fed stream 0's input gives stream 0's output
    bo.reset()
    assert torch.equal(run_calls(bo, x, sizes), yo)
    bo.reset()
    y1 = run_calls(bo, x[0:1].expand(S, x.shape[1], C).contiguous(), sizes)
    assert torch.equal(y1[0], yo[0]) and bool((y1 == y1[0:1]).all().item())


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
def test_one_trip_stage_feeds_the_next_convolver(amd, tmp_path):
    """BASELINE config 5's shape in small: `hilbert -p 4095` (one-trip) writing the pair rings of a long fir_p behind it (the fp64 stand-in for the
    zita_convolver stage, which the reference build here cannot check), without the LTI merge that would make one filter of the two"""
    import torch
    f = os.path.join(str(tmp_path), "h.raw")
    np.asarray(make_filter(20000, seed=3, decay=3000.0), dtype="<f8").tofile(f)
    chain = f"hilbert -p 4095 fir_p -t pcm -e double -c 1 {f}"
    S, C, sizes = 24, 2, (65536, 65536, 20000)
    os.environ["DSP_AMD_NO_LTI_MERGE"] = "1"
    try:
        bo, bs = build(amd, chain, C, S, max(sizes), True), build(amd, chain, C, S, max(sizes), False)
    finally:
        os.environ.pop("DSP_AMD_NO_LTI_MERGE")
    assert "one-trip" in bo.plan() and "fed-by-conv" in bo.plan() and "N=16384=" in bo.plan(), bo.plan()
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x = torch.rand((S, sum(sizes), C), dtype=torch.float64, device="cuda", generator=g) - 0.5
    yo, ys = run_calls(bo, x, sizes), run_calls(bs, x, sizes)
    assert yo.shape == ys.shape and float((yo - ys).pow(2).mean().sqrt()) < 1e-13
    for s in (0, 11, 23):
        ref = RefChain(chain, 48000, C).process(x[s].cpu().numpy(), block=4096)
        got = yo[s].cpu().numpy()
        assert ref.shape == got.shape and rms(ref - got) < 1e-12, (s, ref.shape, got.shape, rms(ref - got))


def test_where_the_one_trip_form_is_not_used(amd, tmp_path):
    """longer filters, short calls (the small-call regime), the float32 contract and resamplers keep their plans"""
    f = os.path.join(str(tmp_path), "h.raw")
    np.asarray(make_filter(5000), dtype="<f8").tofile(f)
    g = os.path.join(str(tmp_path), "g.raw")
    np.asarray(make_filter(3000), dtype="<f8").tofile(g)
    assert "one-trip" not in amd.BatchChain(f"fir_p -t pcm -e double -c 1 {f}", 48000, 2, 4, 16384).plan()          # 5000 taps, calls that do not fill the larger window's blocks
    assert "one-trip" in amd.BatchChain(f"fir_p -t pcm -e double -c 1 {f}", 48000, 2, 4, 65536).plan()              # ... and calls that do
    h9 = os.path.join(str(tmp_path), "h9.raw")
    np.asarray(make_filter(8194), dtype="<f8").tofile(h9)
    assert "one-trip" not in amd.BatchChain(f"fir_p -t pcm -e double -c 1 {h9}", 48000, 2, 4, 65536).plan()         # 8194 taps
    assert "one-trip" not in amd.BatchChain(f"fir_p -t pcm -e double -c 1 {g}", 48000, 2, 4, 256).plan()            # 256-frame calls
    assert "one-trip" not in amd.BatchChain(f"zita_convolver -t pcm -e double -c 1 {g}", 48000, 2, 4, 16384).plan()
    assert "one-trip" not in amd.BatchChain(f"fir_p -t pcm -e double -c 1 {g} resample 96k", 48000, 2, 4, 16384).plan()
    assert "one-trip" in amd.BatchChain(f"fir_p -t pcm -e double -c 1 {g}", 48000, 2, 4, 16384).plan()
