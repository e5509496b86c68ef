"""Seeded random effects chains through the stand-alone host path (`EffectsChain`: chain parser -> merge/optimise -> fused
device pipeline) against the REAL reference's chain runtime in-process (oracle/_ref/libdspref.so) on the same input, with
different call sizes on the two sides.  The generator draws from every effect this library provides -- all biquad types and
width units, time-reversed sections, gains, remix, integer / fractional delays, direct and FFT FIRs, hilbert, resample,
mid/side, crossfeed -- under random channel selections; what the reference refuses must be refused here too."""
import numpy as np
import pytest

from oracle_api import RefChain, rms

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")]


@pytest.fixture(scope="module")
def amd():
    import dsp_amd
    assert dsp_amd.load_library().dspamd_device_count() >= 1, "no HIP device: GPU tests must fail loudly, not fall back"
    return dsp_amd


def gen_chain(rng, channels, cascade_only=False, extended=False):
    """-> chain text; cascade_only: sections and gains under channel selections, nothing else; extended: more exotic arguments
    (odd resampling ratios and bandwidths, one filter per channel, hilbert angles, high all-pass orders)"""
    ch, fs, out = channels, 48000, []

    def f0(lo=30.0, hi=9000.0):
        return "%.6g" % float(np.exp(rng.uniform(np.log(lo), np.log(min(hi, 0.4 * fs)))))

    def width(shelf=False):
        k = rng.integers(0, 5 if shelf else 4)
        if k == 0: return "%.4g" % rng.uniform(0.3, 4.0)
        if k == 1: return "%.4gq" % rng.uniform(0.3, 4.0)
        if k == 2: return "%.4go" % rng.uniform(0.2, 3.0)
        if k == 3: return "%.4gh" % rng.uniform(20.0, 400.0)
        return "%.4g%s" % ((rng.uniform(0.3, 1.0), "s") if rng.integers(2) else (rng.uniform(3.0, 12.0), "d"))

    def db():
        return "%.4g" % rng.uniform(-9.0, 6.0)

    def coefs(n, decay):
        h = rng.standard_normal(n) * np.exp(-np.arange(n) / decay)
        return "coefs:" + ",".join("%.17g" % v for v in h / np.sqrt(np.sum(h * h)) / 3)

    def selector():
        nonlocal out
        if ch < 2 or rng.integers(3) == 0:
            out.append(":")
            return list(range(ch))
        k = int(rng.integers(1, ch))
        sel = sorted(rng.choice(ch, size=k, replace=False).tolist())
        out.append(":" + ",".join(str(c) for c in sel))
        return sel

    n_sel = ch
    for _ in range(int(rng.integers(2, 9))):
        if rng.integers(4) == 0:
            n_sel = len(selector())
        r = rng.integers(0, 52) if cascade_only else rng.integers(0, 100)
        if cascade_only and 11 <= r < 13: r = 0            # (`add` is not linear in the state: the chunked cascade steps aside)
        rev = " -r" if rng.integers(8) == 0 and not cascade_only else ""
        if r < 8: out.append(f"gain {db()}")
        elif r < 11: out.append("mult %.5g" % rng.uniform(-1.5, 1.5))
        elif r < 13: out.append("add %.3g" % rng.uniform(-1e-3, 1e-3))
        elif r < 22: out.append(f"{rng.choice(['lowpass', 'highpass', 'bandpass_skirt', 'bandpass_peak', 'notch', 'allpass'])}{rev} {f0()} {width()}")
        elif r < 32: out.append(f"eq{rev} {f0()} {width()} {db()}")
        elif r < 38: out.append(f"{rng.choice(['lowshelf', 'highshelf'])}{rev} {f0(80, 6000)} {width(True)} {db()}")
        elif r < 43: out.append(f"{rng.choice(['lowpass_1', 'highpass_1', 'allpass_1', 'lowpass_1p'])} {f0()}")
        elif r < 46: out.append(f"{rng.choice(['lowshelf_1', 'highshelf_1'])} {f0()} {db()}")
        elif r < 48: out.append(f"linkwitz_transform {f0(40, 120)} 0.9 {f0(20, 60)} 0.6")
        elif r < 50: out.append("deemph")
        elif r < 52: out.append("biquad 0.5 0.2 -0.1 1.0 -0.3 0.2")
        elif r < 58:
            out.append("delay %dS" % rng.integers(0, 300) if rng.integers(3) else "delay %.4gm" % rng.uniform(0.05, 3.0))
        elif r < 63: out.append("delay -f%s %.5gS" % (rng.choice(["", "1", "2", "5", "9", "16"] if extended else ["", "1", "2", "5"]), rng.uniform(0.05, 40.0)))
        elif r < 69: out.append(f"fir {coefs(int(rng.integers(2, 33)), 6.0)}")                # direct form
        elif r < 74:
            if extended and n_sel > 1 and rng.integers(2):                                     # one filter per selected channel, ragged lengths
                per = "/".join(coefs(int(rng.integers(3, 200)), 40.0)[6:] for _ in range(n_sel))
                out.append(f"{rng.choice(['fir', 'fir_p'])} coefs:{per}")
            else:
                out.append(f"{rng.choice(['fir', 'fir_p'])} {coefs(int(rng.integers(40, 700)), 90.0)}")
        elif r < 77:
            opt = rng.choice(["", " -p", " -a 45", " -p -a -30"]) if extended else rng.choice(["", " -p"])
            out.append("hilbert%s %d" % (opt, 2 * int(rng.integers(20, 300)) + 1))
        elif r < 82 and fs == 48000:
            new = int(rng.choice([44100, 96000, 24000, 32000, 16000, 88200, 37800, 22050, 47999] if extended else [44100, 96000, 24000, 32000]))
            out.append(":"); n_sel = ch
            out.append(f"resample {'0.9 ' if extended and rng.integers(3) == 0 else ''}{new}")
            fs = new
        elif r < 88:
            out.append(":"); n_sel = ch
            k = int(rng.integers(1, 5))
            rows = []
            for _ in range(k):
                m = int(rng.integers(0, min(ch, 3) + 1))
                rows.append(",".join(str(c) for c in sorted(rng.choice(ch, size=m, replace=False).tolist())) if m else ".")
            out.append("remix " + " ".join(rows))
            ch = n_sel = k
        elif r < 94 and n_sel == 2:
            out.append(str(rng.choice(["st2ms", "ms2st"])))
        elif r < 100 and n_sel == 2:
            out.append("crossfeed %s %.3g" % (f0(300, 1200), rng.uniform(2.0, 9.0)))
        else:
            out.append(f"eq {f0()} 1.0 {db()}")
    return " ".join(out)


@pytest.mark.parametrize("seed", range(240))
def test_random_chain_vs_real_reference(amd, seed):
    rng = np.random.Generator(np.random.PCG64(7000 + seed))
    channels = int(rng.choice([1, 2, 2, 3, 4, 6]))
    chain = gen_chain(rng, channels)
    n = int(rng.integers(3000, 20000))
    x = rng.uniform(-0.5, 0.5, size=(n, channels))
    try:
        ref = RefChain(chain, 48000, channels)
    except ValueError:
        with pytest.raises(ValueError):
            amd.EffectsChain(chain, 48000, channels)
        return
    yr = ref.process(x, block=int(rng.choice([512, 2048, 4096])))
    ec = amd.EffectsChain(chain, 48000, channels)
    y = ec.process(x, block=int(rng.choice([333, 1024, 2048, 5000])))
    assert (ec.ofs, ec.ochannels) == (ref.ofs, ref.ochannels), chain
    assert y.shape == yr.shape, (chain, y.shape, yr.shape)
    scale = max(rms(yr), 1e-3)
    assert rms(y - yr) <= 1e-10 * scale, (chain, rms(y - yr), scale)


@pytest.mark.parametrize("seed", range(60))
def test_random_chain_batch_vs_real_reference(amd, seed):
    # the same generator through the BATCH path (S independent streams in one fused pipeline, buffers resident in HBM): many
    # streams reach the kernels that share the chip differently (cascade_rows / _wave / _fast, chunked cascade, batched FFTs)
    import torch
    rng = np.random.Generator(np.random.PCG64(11000 + seed))
    channels = int(rng.choice([2, 4, 8]))
    S = int(rng.choice([3, 40, 130, 260]))
    chain = gen_chain(rng, channels)
    block = int(rng.choice([2048, 4096, 8192]))
    n = block * int(rng.integers(1, 4)) + int(rng.integers(0, block))
    x = rng.uniform(-0.5, 0.5, size=(S, n, channels))
    try:
        RefChain(chain, 48000, channels)
    except ValueError:
        with pytest.raises(ValueError):
            amd.BatchChain(chain, 48000, channels, S, block)
        return
    b = amd.BatchChain(chain, 48000, channels, S, block)
    y = b.process(torch.from_numpy(x).cuda(), block).cpu().numpy()
    for s in sorted({0, S // 2, S - 1}):
        yr = RefChain(chain, 48000, channels).process(x[s], block=2048)
        assert y[s].shape == yr.shape, (chain, s, y[s].shape, yr.shape)
        assert rms(y[s] - yr) <= 1e-10 * max(rms(yr), 1e-3), (chain, s, rms(y[s] - yr))


@pytest.mark.parametrize("seed", range(40))
def test_random_cascade_chunked_vs_real_reference(amd, monkeypatch, seed):
    # few channels, long calls, sections and gains only: the chunked cascade (kernels_chunk.hip) on random section sets,
    # channel selections (several table classes) and call sizes; three calls, so the carried state crosses call boundaries
    import torch
    monkeypatch.setenv("DSP_AMD_CASCADE_CHUNKS", "4096")      # always chunk (by default only where the cost model says it pays)
    rng = np.random.Generator(np.random.PCG64(13000 + seed))
    channels = int(rng.choice([1, 2, 4, 8]))
    S = int(rng.choice([1, 1, 2, 5]))
    chain = gen_chain(rng, channels, cascade_only=True)
    block = 2048 * int(rng.integers(4, 33))               # (whole tiles of every cascade kernel: a chunk plan always exists)
    x = rng.uniform(-0.5, 0.5, size=(S, 3 * block, channels))
    b = amd.BatchChain(chain, 48000, channels, S, block)
    L = amd.load_library()
    L.dspamd_profile_enable(1)
    y = b.process(torch.from_numpy(x).cuda(), block).cpu().numpy()
    names = {ln.split()[0] for ln in L.dspamd_profile_collect().decode().splitlines()}
    L.dspamd_profile_enable(0)
    if any(t[0].isalpha() and t not in ("gain", "mult") for t in chain.split()):     # at least one section
        assert "cascade_chunk_fix" in names, (chain, block, names)
    for s in sorted({0, S - 1}):
        yr = RefChain(chain, 48000, channels).process(x[s], block=2048)
        assert y[s].shape == yr.shape, (chain, s, y[s].shape, yr.shape)
        assert rms(y[s] - yr) <= 1e-10 * max(rms(yr), 1e-3), (chain, s, block, rms(y[s] - yr))


@pytest.mark.parametrize("seed", range(120))
def test_random_chain_extended_vs_real_reference(amd, seed):
    # more exotic arguments: odd resampling ratios (47999: 47999/48000) and bandwidths, one filter per selected channel with
    # ragged lengths, hilbert phase angles, all-pass orders up to 16
    rng = np.random.Generator(np.random.PCG64(17000 + seed))
    channels = int(rng.choice([1, 2, 2, 3, 4]))
    chain = gen_chain(rng, channels, extended=True)
    x = rng.uniform(-0.5, 0.5, size=(int(rng.integers(3000, 12000)), channels))
    try:
        ref = RefChain(chain, 48000, channels)
    except ValueError:
        with pytest.raises(ValueError):
            amd.EffectsChain(chain, 48000, channels)
        return
    yr = ref.process(x, block=int(rng.choice([512, 2048, 4096])))
    ec = amd.EffectsChain(chain, 48000, channels)
    y = ec.process(x, block=int(rng.choice([333, 1024, 2048, 5000])))
    assert (ec.ofs, ec.ochannels) == (ref.ofs, ref.ochannels), chain
    assert y.shape == yr.shape, (chain[:300], y.shape, yr.shape)
    assert rms(y - yr) <= 1e-10 * max(rms(yr), 1e-3), (chain[:300], rms(y - yr))


@pytest.mark.parametrize("seed", range(60))
def test_random_chain_ragged_calls(amd, seed):
    # calls of any size, one after the other (a LADSPA host's 1 ... 1024 frames, then a big one): the stream the calls add up
    # to must not depend on how it was cut -- carried states, convolver rings, resampler phases, mapped-staging vs copy path
    rng = np.random.Generator(np.random.PCG64(19000 + seed))
    channels = int(rng.choice([1, 2, 2, 3, 4]))
    chain = gen_chain(rng, channels, extended=bool(seed & 1))
    n = int(rng.integers(4000, 16000))
    x = rng.uniform(-0.5, 0.5, size=(n, channels))
    try:
        ref = RefChain(chain, 48000, channels)
    except ValueError:
        with pytest.raises(ValueError):
            amd.EffectsChain(chain, 48000, channels)
        return
    yr = ref.process(x, block=2048)
    ec = amd.EffectsChain(chain, 48000, channels)
    outs, p = [], 0
    while p < n:
        k = int(rng.choice([1, 2, 7, 64, 100, 256, 1000, 1024, 3000, 5000]))
        outs.append(ec.run(x[p:p + k]))
        p += k
    while True:
        o = ec.drain(2048)
        if o is None:
            break
        outs.append(o)
    y = np.concatenate([o for o in outs if o.shape[0]]) if any(o.shape[0] for o in outs) else np.zeros((0, ec.ochannels))
    assert y.shape == yr.shape, (chain[:300], y.shape, yr.shape)
    assert rms(y - yr) <= 1e-10 * max(rms(yr), 1e-3), (chain[:300], rms(y - yr))


@pytest.mark.parametrize("seed", range(30))
def test_random_chain_small_calls_vs_real_reference(amd, tmp_path, seed):
    # the convolver's small-call regime (delay-line head + overlap-save tail, DESIGN.md 4.3, docs/history.md 4.2b) inside random chains: a long
    # zero-latency filter on every channel between randomly drawn cascade effects, fed in calls of 256 ... 2048 frames that
    # cross several tail hand-overs, with a ragged last call (which takes the stream off the grid) and the drain
    import torch
    rng = np.random.Generator(np.random.PCG64(17000 + seed))
    channels = int(rng.choice([1, 2, 3, 4, 8]))
    S = int(rng.choice([1, 5, 64]))
    block = int(rng.choice([256, 512, 1024, 2048]))
    taps = int(rng.integers(8 * block, 24 * block))
    h = rng.standard_normal(taps) * np.exp(-np.arange(taps) / (taps / 7.0))
    h = h / np.sqrt(np.sum(h * h)) / 3.0
    p = str(tmp_path / "h.raw")
    np.asarray(h, dtype="<f8").tofile(p)
    head = gen_chain(rng, channels, cascade_only=True) if rng.integers(3) else ""
    tail = gen_chain(rng, channels, cascade_only=True) if rng.integers(3) == 0 else ""
    chain = f"{head} : fir_p -t pcm -e double -c 1 {p} {tail}".strip()
    n = block * int(rng.integers(9, 26)) + int(rng.integers(0, block))
    x = rng.uniform(-0.5, 0.5, size=(S, n, channels))
    b = amd.BatchChain(chain, 48000, channels, S, block)
    assert "small-calls" in b.plan(), b.plan()
    y = b.process(torch.from_numpy(x).cuda(), block).cpu().numpy()
    for s in sorted({0, S - 1}):
        yr = RefChain(chain, 48000, channels).process(x[s], block=2048)
        assert y[s].shape == yr.shape, (chain, s, y[s].shape, yr.shape)
        assert rms(y[s] - yr) <= 1e-10 * max(rms(yr), 1e-3), (chain, s, rms(y[s] - yr))
