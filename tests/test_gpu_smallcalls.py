"""GPU parity of the small-call regime of the FFT convolver (calls much shorter than the filter: the reference's own
block is 2048 frames, dsp.h:38; its fir_p serves them with a partitioned delay line, fir_p.c:64-103): head partitions
through the delay-line kernel + the rest of the filter through the overlap-save convolver once per 8 blocks, against the
real reference at the same block size, across the boundaries where the tail hands over, off-grid calls, reset, drain."""
import os

import numpy as np
import pytest

from oracle_api import RefChain, rms

pytestmark = pytest.mark.gpu


def noise(frames, ch, seed, amp=0.5):
    return np.random.Generator(np.random.PCG64(seed)).uniform(-amp, amp, size=(frames, ch))


@pytest.fixture(scope="module")
def amd():
    import dsp_amd
    assert dsp_amd.load_library().dspamd_device_count() >= 1, "no HIP device: GPU tests must fail loudly, not fall back"
    return dsp_amd


def filt(tmp_path, taps, seed=7, name="h.raw"):
    rng = np.random.default_rng(seed)
    h = rng.standard_normal(taps) * np.exp(-np.arange(taps) / (taps / 8.0))
    h = h / np.sqrt(np.sum(h * h)) / 4.0
    p = os.path.join(str(tmp_path), name)
    np.asarray(h, dtype="<f8").tofile(p)
    return p, h


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
@pytest.mark.parametrize("taps,block,S,C,chain_head", [
    (65536, 2048, 3, 8, "lowpass 1k 0.707 eq 400 2.0 1.5 "),      # the headline chain's shape at the reference's block size: head 8 x 2048 + tail
    (65536, 2048, 2, 2, ""),                                       # convolver first in the chain (de-interleaving pass feeds the rings)
    (100000, 2048, 2, 2, "gain -2 "),                              # six tail partitions of 16384 taps (the last one ragged): the tail's delay line wraps
    (20000, 1024, 2, 4, "gain -3 "),                               # 1024-frame partitions, tail of 11808 taps
    (9000, 512, 3, 2, ""),                                         # short enough for the delay line alone (18 partitions -> no: 9000/512 = 18 > 16, tail)
    (7000, 512, 2, 3, "highpass 50 0.707 "),                       # 14 partitions, no tail; odd channel count (a half-empty pair)
    (60000, 4096, 2, 2, ""),                                       # calls of two partitions each
    (40000, 2048, 2, 4, "FIR"),                                    # `fir`: the same values its latency late -- the head's and the tail's windows start that much earlier
    (5000, 512, 2, 2, "FIR"),                                      # ... the whole filter in the head's delay line
])
def test_small_calls_vs_real_reference(amd, tmp_path, taps, block, S, C, chain_head):
    import torch
    p, h = filt(tmp_path, taps)
    chain = f"{chain_head}fir_p -t pcm -e double -c 1 {p}" if chain_head != "FIR" else f"fir -t pcm -e double -c 1 {p}"
    n_blocks = 2 * 8 + 3 if taps > 16 * min(block, 2048) else 12      # across two tail hand-overs
    if taps >= 65536:
        n_blocks = (5 if taps == 65536 else 8) * 8 + 3                # ... and once around the tail's own delay line (3 / 6 partitions)
    N = n_blocks * block
    x = np.stack([noise(N, C, 300 + s) for s in range(S)])
    b = amd.BatchChain(chain, 48000, C, S, block)
    assert "small-calls" in b.plan(), b.plan()
    if taps >= 65536 and not os.environ.get("DSP_AMD_CONV_UPC"):
        assert "+ tail " in b.plan() and "taps delay line N=16384" in b.plan(), b.plan()   # the tail as slots of 8192 taps through 16384-point transforms
    y = b.process(torch.from_numpy(x).cuda(), block).cpu().numpy()
    for s in range(S):
        ref = RefChain(chain, 48000, C).process(x[s], block=block)
        assert y[s].shape == ref.shape, (y[s].shape, ref.shape)
        assert rms(y[s] - ref) < 1e-12, (s, rms(y[s] - ref))


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
@pytest.mark.parametrize("taps,block,S,C,chain_head", [
    (65536, 8192, 2, 8, "lowpass 1k 0.707 eq 400 2.0 1.5 "),      # the headline chain at 8192-frame calls: 8 slots, 16384-point transforms, rings fed by the cascade
    (65536, 16384, 2, 2, ""),                                      # convolver first in the chain (de-interleaving pass feeds the rings), 4 slots
    (40000, 4096, 2, 2, ""),                                       # 10 slots of 4096 taps, the last one ragged; the smallest transform (8192 points)
    (100000, 32768, 1, 3, "gain -2 "),                             # 4 slots at 65536-point transforms; odd channel count (a half-empty pair)
    (16384, 8192, 2, 4, ""),                                       # the shortest filter the regime takes: two whole slots
    (40000, 12288, 2, 2, "gain -1 "),                              # calls of three blocks of 4096 frames
    (30000, 8192, 2, 8, "FIR"),                                    # `fir` (the same values, its latency late): the child's windows start that much earlier, the frames the chain discards are dropped by K3
    (20000, 8192, 3, 4, ":0,2 "),                                  # two of four channels selected: the others pass through the de-interleaving pass
])
def test_mid_size_calls_vs_real_reference(amd, tmp_path, taps, block, S, C, chain_head):
    """calls of a power of two of frames between 4096 and half the filter: the whole filter as slots of `block` taps in the row
    kernel's delay line (conv_row mode 3), twice around the delay line, against the real reference at the same block size"""
    import torch
    p, h = filt(tmp_path, taps)
    chain = f"{chain_head}fir_p -t pcm -e double -c 1 {p}" if chain_head != "FIR" else f"fir -t pcm -e double -c 1 {p}"
    F = block & -block                                            # the regime's block: the largest power of two that divides the call size
    P = -(-taps // F)
    n_blocks = (2 * P + 2) * F // block + 1
    N = n_blocks * block
    x = np.stack([noise(N, C, 500 + s) for s in range(S)])
    b = amd.BatchChain(chain, 48000, C, S, block)
    if os.environ.get("DSP_AMD_CONV_UPC") == "0":
        assert "mid-size-calls" not in b.plan(), b.plan()
    else:
        assert f"mid-size-calls: {P}x{F} taps delay line N={2 * F}" in b.plan(), b.plan()
    y = b.process(torch.from_numpy(x).cuda(), block).cpu().numpy()
    for s in range(S):
        ref = RefChain(chain, 48000, C).process(x[s], block=block)
        assert y[s].shape == ref.shape, (y[s].shape, ref.shape)
        assert rms(y[s] - ref) < 1e-12, (s, rms(y[s] - ref))


def filt_channels(tmp_path, taps, n, name="hn.raw"):
    """a filter file of n channels: one filter per selected channel, in channel order (fir_p.c:483-495, fir.c:342-357)"""
    hs = []
    for c in range(n):
        rng = np.random.default_rng(70 + c)
        h = rng.standard_normal(taps) * np.exp(-np.arange(taps) / (taps / (6.0 + c)))
        hs.append(h / np.sqrt(np.sum(h * h)) / 4.0)
    p = os.path.join(str(tmp_path), name)
    np.asarray(np.stack(hs, axis=1), dtype="<f8").tofile(p)
    return p


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
@pytest.mark.parametrize("taps,block,S,C,sel,regime", [
    (65536, 2048, 2, 4, "", "small-calls"),                       # four filters of the headline's length at the reference's block: head 4 x 2048 + delay-line tail, a filter per pair
    (20000, 1024, 2, 3, "", "small-calls"),                       # odd channel count, plain tail
    (7000, 512, 2, 4, ":1,3 ", "small-calls"),                    # two of four channels, two filters: the whole filter in the head's delay line
    (65536, 8192, 2, 4, "", "mid-size-calls"),                    # the row kernel's delay line with a row of partition spectra per filter
    (40000, 4096, 2, 2, "", "mid-size-calls"),
    (30000, 8192, 2, 5, ":0,2,4 ", "mid-size-calls"),
])
def test_one_filter_per_channel_in_the_call_size_regimes(amd, tmp_path, taps, block, S, C, sel, regime):
    """round 4: a filter file with one filter per selected channel (fir_p.c:483-495) in the small-call and mid-size-call regimes (round 3 sent it
    down the one-transform-per-call plan): every pair carries one channel and reads its own filter's partition spectra"""
    import torch
    n_sel = C if not sel else len(sel.strip(": ").split(","))
    p = filt_channels(tmp_path, taps, n_sel)
    chain = f"{sel}fir_p -t pcm -e double -c {n_sel} {p}"
    F = block & -block
    n_blocks = (2 * (-(-taps // F)) + 2) * F // block + 1 if regime == "mid-size-calls" else 2 * 8 + 3
    if taps >= 65536 and regime == "small-calls": n_blocks = 5 * 8 + 3
    N = n_blocks * block
    x = np.stack([noise(N, C, 700 + s) for s in range(S)])
    b = amd.BatchChain(chain, 48000, C, S, block)
    assert regime in b.plan() and "per-channel-filters" in b.plan(), b.plan()
    y = b.process(torch.from_numpy(x).cuda(), block).cpu().numpy()
    for s in range(S):
        ref = RefChain(chain, 48000, C).process(x[s], block=block)
        assert y[s].shape == ref.shape, (y[s].shape, ref.shape)
        assert rms(y[s] - ref) < 1e-12, (s, rms(y[s] - ref))


def test_mid_size_calls_off_the_grid_and_reset(amd, tmp_path):
    """a stream that leaves the grid (a short call) carries on from the rings with one transform per call; reset() returns it
    to the delay-line path; both equal a batch created for long calls"""
    import torch
    p, h = filt(tmp_path, 30000)
    chain = f"eq 300 1.0 3 fir_p -t pcm -e double -c 1 {p}"
    S, C, block = 2, 4, 8192
    x = torch.from_numpy(np.stack([noise(9 * block + 700, C, 40 + s) for s in range(S)])).cuda()
    mid = amd.BatchChain(chain, 48000, C, S, block)
    big = amd.BatchChain(chain, 48000, C, S, 1 << 17)
    assert "mid-size-calls" in mid.plan() and "mid-size-calls" not in big.plan() and "small-calls" not in mid.plan()
    y_big = big.run(x).clone()

    def in_calls(sizes):
        outs, pos = [], 0
        for n in sizes:
            outs.append(mid.run(x[:, pos:pos + n, :].contiguous()).clone())
            pos += n
        assert pos == x.shape[1]
        return torch.cat(outs, dim=1)

    y = in_calls([block] * 5 + [700] + [block] * 4)
    assert float((y - y_big).abs().max()) < 1e-12
    mid.reset()
    y = in_calls([block] * 9 + [700])                       # on the grid all the way after the reset
    assert float((y - y_big).abs().max()) < 1e-12
    mid.reset()
    y = in_calls([5000, 3192] + [block] * 8 + [700])        # never on the grid's block size at position 0 ... and a later start on the grid is not taken up
    assert float((y - y_big).abs().max()) < 1e-12


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
@pytest.mark.parametrize("seed", range(int(os.environ.get("DSPAMD_FUZZ_CALL_SIZES", "16"))))     # (more seeds: DSPAMD_FUZZ_CALL_SIZES=400)
def test_random_call_sizes_vs_real_reference(amd, tmp_path, seed):
    """filters of 3000 ... 90000 taps at call sizes from 512 to 24576 frames, `fir` and `fir_p`, with effects in front of and behind the
    convolver, a ragged call in mid-stream for every second seed: whichever regime the planner picks (delay-line head + tail, the
    whole filter as delay-line slots, one transform per call) and wherever the stream leaves it, the samples are the reference's"""
    import torch
    rng = np.random.default_rng(1000 + seed)
    taps = int(rng.choice([3000, 9000, 20000, 33000, 50000, 70000, 90000]))
    block = int(rng.choice([512, 1024, 2048, 4096, 6144, 8192, 12288, 16384, 24576]))
    S, C = int(rng.integers(1, 4)), int(rng.integers(1, 6))
    eff = str(rng.choice(["fir_p", "fir"]))
    head = str(rng.choice(["", "gain -2 ", "eq 500 1.0 3 lowpass 8k 0.7 "]))
    tail = str(rng.choice(["", " gain 1.5", " eq 2k 1.0 -3"]))
    if seed % 3 == 2 and C > 1:
        # (round 4: every third seed with one filter per channel, fir_p.c:483-495 -- the same regimes, a pair per channel)
        p = filt_channels(tmp_path, taps, C, name=f"hn{seed}.raw")
        chain = f"{head}{eff} -t pcm -e double -c {C} {p}{tail}"
    else:
        p, h = filt(tmp_path, taps, seed=seed)
        chain = f"{head}{eff} -t pcm -e double -c 1 {p}{tail}"
    n_calls = min(max(3, (2 * taps) // block + 3), 40)
    sizes = [block] * n_calls
    if seed % 2:
        sizes.insert(n_calls // 2, int(rng.integers(1, block)))
    N = sum(sizes)
    x = np.stack([noise(N, C, 7000 + 10 * seed + s) for s in range(S)])
    b = amd.BatchChain(chain, 48000, C, S, block)
    xt = torch.from_numpy(x).cuda()
    outs, pos = [], 0
    for n in sizes:
        outs.append(b.run(xt[:, pos:pos + n, :].contiguous()).clone())
        pos += n
    while True:
        y = b.drain(block)
        if y is None:
            break
        outs.append(y.clone())
    y = torch.cat(outs, dim=1).cpu().numpy()
    for s in range(S):
        # (the reference in calls of `block` frames throughout: the stream these effects make does not depend on how it is cut into calls)
        ref = RefChain(chain, 48000, C).process(x[s], block=block)
        assert y[s].shape == ref.shape, (b.plan(), y[s].shape, ref.shape)
        assert rms(y[s] - ref) < 1e-12, (b.plan(), s, rms(y[s] - ref))


def test_small_calls_equal_one_transform_per_call(amd, tmp_path):
    """the same stream through the small-call path and through the one-transform-per-call path (a batch created for large
    calls), and a stream that leaves the grid half way (an odd call size) and continues on the rings"""
    import torch
    p, h = filt(tmp_path, 30000)
    chain = f"eq 300 1.0 3 fir_p -t pcm -e double -c 1 {p}"
    S, C, block = 2, 4, 1024
    x = torch.from_numpy(np.stack([noise(24 * block + 700, C, 20 + s) for s in range(S)])).cuda()
    small = amd.BatchChain(chain, 48000, C, S, block)
    big = amd.BatchChain(chain, 48000, C, S, 1 << 16)
    assert "small-calls" in small.plan() and "small-calls" not in big.plan()
    y_big = big.run(x).clone()
    outs = [small.run(x[:, q:q + block, :].contiguous()).clone() for q in range(0, 11 * block, block)]
    outs.append(small.run(x[:, 11 * block:11 * block + 700, :].contiguous()).clone())          # off the grid: 700 frames
    pos = 11 * block + 700
    while pos < x.shape[1]:
        n = min(block, x.shape[1] - pos)
        outs.append(small.run(x[:, pos:pos + n, :].contiguous()).clone())
        pos += n
    y = torch.cat(outs, dim=1)
    assert y.shape == y_big.shape
    assert float((y - y_big).abs().max()) < 1e-12


def test_small_calls_reset(amd, tmp_path):
    import torch
    for taps, nb in ((40000, 11), (70000, 27)):          # an overlap-save tail; a tail with a delay line of its own (state in three places)
        p, h = filt(tmp_path, taps, name=f"h{taps}.raw")
        chain = f"fir_p -t pcm -e double -c 1 {p}"
        S, C, block = 2, 2, 2048
        x1 = torch.from_numpy(np.stack([noise(nb * block, C, 1 + s) for s in range(S)])).cuda()
        x2 = torch.from_numpy(np.stack([noise(nb * block, C, 9 + s) for s in range(S)])).cuda()
        b = amd.BatchChain(chain, 48000, C, S, block)
        for q in range(0, nb * block, block):
            b.run(x1[:, q:q + block, :].contiguous())
        b.reset()
        y = torch.cat([b.run(x2[:, q:q + block, :].contiguous()).clone() for q in range(0, nb * block, block)], dim=1)
        f = amd.BatchChain(chain, 48000, C, S, block)
        yf = torch.cat([f.run(x2[:, q:q + block, :].contiguous()).clone() for q in range(0, nb * block, block)], dim=1)
        assert torch.equal(y, yf), taps


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
def test_opt_in_iir_merge_vs_real_reference(amd, tmp_path, monkeypatch):
    """DSP_AMD_MERGE_IIR=1 (off by default): sections and gains on every channel in front of a zero-latency convolution are folded
    into the filter once their joint impulse response has decayed below 2^-70 of its peak -- same stream as the reference's
    biquads + fir_p (<= 1e-12 RMS incl. the drain), no cascade stage in the plan; a chain whose poles sit too close to the
    unit circle (2 Hz high-pass) keeps its cascade."""
    import torch
    p, h = filt(tmp_path, 6000)
    biquads = "gain -2 lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 800 1.0 -1 highpass 20 0.707"
    chain = f"{biquads} fir_p -t pcm -e double -c 1 {p}"
    S, C, block = 2, 4, 8192
    x = np.stack([noise(5 * block, C, 60 + s) for s in range(S)])
    monkeypatch.setenv("DSP_AMD_MERGE_IIR", "1")
    b = amd.BatchChain(chain, 48000, C, S, block)
    assert "cascade[" not in b.plan() and "highpass+fir_p" in b.plan(), b.plan()
    y = b.process(torch.from_numpy(x).cuda(), block).cpu().numpy()
    for s in range(S):
        ref = RefChain(chain, 48000, C).process(x[s], block=2048)
        assert y[s].shape == ref.shape and rms(y[s] - ref) < 1e-12, rms(y[s] - ref)
    slow = amd.BatchChain(f"highpass 2 0.707 fir_p -t pcm -e double -c 1 {p}", 48000, C, S, block)
    assert "cascade[" in slow.plan()
    monkeypatch.delenv("DSP_AMD_MERGE_IIR")
    assert "cascade[" in amd.BatchChain(chain, 48000, C, S, block).plan()
