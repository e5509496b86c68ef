"""The cascade fused into the FFT convolver's first pass (kernels_fused.hip: fused_prepass, cascade_chunk_carry, fused_col_fwd) --
the path the headline workload takes -- against the real reference and against the separate kernels (DSP_AMD_FUSE=0), at shapes the
CPU finishes in seconds: every instance (section counts padded to 1 / 2 / 4 / 6 / 8 / 10 / 12, history of 16 and 32 rows, row
segments of 1 / 2 / 4), gains folded in, calls that leave the grid and come back, reset, padded slabs.  The headline's own shape:
tests/test_gpu_conv.py::test_bench_default_configuration_full_size_vs_real_reference."""
import os

import numpy as np
import pytest

from oracle_api import RefChain, rms

pytestmark = pytest.mark.gpu

SECTIONS = ["lowpass 1k 0.707", "highshelf 8k 0.7 -3", "eq 100 1.0 3", "eq 200 1.0 -2", "eq 400 2.0 1.5", "eq 800 1.0 -1", "eq 1600 1.4 2",
            "eq 3200 1.0 -2.5", "eq 6400 3.0 1", "highpass 20 0.707", "lowshelf 150 0.8 2", "eq 5000 2.0 -1.5"]


def make_filter(n, seed=7, decay=8000.0):
    rng = np.random.default_rng(seed)
    h = rng.standard_normal(n) * np.exp(-np.arange(n) / decay)
    return h / np.sqrt(np.sum(h * h)) / 4.0


@pytest.fixture(scope="module")
def amd():
    import dsp_amd
    assert dsp_amd.load_library().dspamd_device_count() >= 1
    return dsp_amd


def build(amd, chain, C, S, B, fuse):
    os.environ["DSP_AMD_FUSE"] = "1" if fuse else "0"
    try:
        return amd.BatchChain(chain, 48000, C, S, B)
    finally:
        os.environ.pop("DSP_AMD_FUSE")


# (sections, gains, taps, hop, S, C, chunks in the plan): N = 2^18 = 256 x 1024 (hop = N - taps) except the last shape, the headline's own N = 2^20 = 256 x 4096
SHAPES = [
    (10, False, 16384, 245760, 8, 8, "960 chunks of 256"),      # the headline's ten sections, 16 history rows, four segments per row
    (1, True, 16384, 245760, 2, 4, "960 chunks of 256"),        # one section between gains, four segments
    (3, True, 32768, 229376, 4, 8, "896 chunks of 256"),        # padded to four sections, 32 history rows
    (5, False, 16384, 245760, 56, 8, "480 chunks of 512"),      # padded to six, two segments
    (7, False, 32768, 229376, 112, 4, "448 chunks of 512"),     # padded to eight; two channel pairs per stream
    (12, False, 16384, 245760, 120, 8, "240 chunks of 1024"),   # twelve sections, whole rows
    (4, False, 11111, 245760, 8, 8, "960 chunks of 256"),       # a filter shorter than 16 rows: the window takes 16 whole rows of history all the same
    (2, False, 20000, 229376, 4, 4, "896 chunks of 256"),       # ... 32 rows
    (10, False, 65536, 983040, 112, 8, "240 chunks of 4096"),   # the headline's own instances: N = 2^20 (256 x 4096), whole rows as chunks (seg = 1), fused_prepass_mm<2>
]


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
@pytest.mark.parametrize("nsec,gains,taps,B,S,C,chunks", SHAPES)
def test_fused_first_pass_vs_reference_and_separate_kernels(amd, tmp_path, nsec, gains, taps, B, S, C, chunks):
    import torch
    f = os.path.join(str(tmp_path), "h.raw")
    np.asarray(make_filter(taps, seed=taps + nsec), dtype="<f8").tofile(f)
    secs = SECTIONS[:nsec]
    if gains:
        secs = ["gain -2.5"] + secs[:1] + ["gain 1.5"] + secs[1:] + ["gain -0.75"]
    chain = " ".join(secs) + f" fir_p -t pcm -e double -c 1 {f}"
    bf, bs = build(amd, chain, C, S, B, True), build(amd, chain, C, S, B, False)
    assert f"cascade-fused({chunks})" in bf.plan(), bf.plan()
    assert "cascade-fused" not in bs.plan()
    g = torch.Generator(device="cuda"); g.manual_seed(1000 + nsec)
    # hop, hop, a call off the grid (separate kernels, on the states and rings the fused ones left), hop again (q_abs a multiple of 8: fused)
    sizes = [B, B, 5000, B]
    pad = 68
    xs = []
    for n in sizes:
        buf = torch.rand((S, n + pad, C), dtype=torch.float64, device="cuda", generator=g) - 0.5
        xs.append(buf[:, :n, :])                       # padded slabs, as bench.py hands them over
    yf = [bf.run(x).clone() for x in xs]
    ys = [bs.run(x).clone() for x in xs]
    for k, (a, b) in enumerate(zip(yf, ys)):
        assert a.shape == b.shape == (S, sizes[k], C)
        d = float((a - b).pow(2).mean().sqrt())
        assert d < 1e-12, (k, d)
    for s in sorted({0, S - 1, (7 * nsec) % S}):
        x = torch.cat([t[s] for t in xs], dim=0).cpu().numpy()
        ref = RefChain(chain, 48000, C).run(x)
        got = torch.cat([t[s] for t in yf], dim=0).cpu().numpy()
        assert ref.shape == got.shape
        assert rms(ref - got) < 1e-12, (s, rms(ref - got))
    # reset: the same call again gives the same bits
    bf.reset()
    assert torch.equal(bf.run(xs[0]), yf[0])
    # every stream given the same input gives the same output
    bf.reset()
    y = bf.run(xs[0][0:1].expand(S, sizes[0], C).contiguous())
    assert torch.equal(y[0], yf[0][0]) and bool((y == y[0:1]).all().item())


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
@pytest.mark.parametrize("taps,S,C,chunks", [(16384, 8, 8, "956 chunks of 256"), (16384, 120, 8, "239 chunks of 1024"), (30000, 6, 4, "904 chunks of 256")])
def test_fused_first_pass_in_front_of_the_2x_resampler(amd, tmp_path, taps, S, C, chunks):
    """BASELINE config 4's shape in small (round 5): sections + fir_p + resample 48k -> 96k = ONE convolver stage with two polyphase branches (fir_p merged
    into the resampler); its calls of one whole hop are whole windows from the second call on (the first drops out_delay outputs), and those take the
    fused first pass -- history of 17 / 30 rows (the run-time-history instance), chunk counts that are no multiple of 8 in the matrix-core prepass.
    Whole streams with the drain against the real reference, call by call against the separate kernels."""
    import torch
    f = os.path.join(str(tmp_path), "h.raw")
    np.asarray(make_filter(taps, seed=taps + 3), dtype="<f8").tofile(f)
    chain = " ".join(SECTIONS[:10]) + f" fir_p -t pcm -e double -c 1 {f} resample 96k"
    rows = -(-(taps + 583 + 7) // 1024)
    B = (1 << 18) - rows * 1024
    bf, bs = build(amd, chain, C, S, B, True), build(amd, chain, C, S, B, False)
    assert f"cascade-fused({chunks})" in bf.plan() and f"hop={B}" in bf.plan(), bf.plan()
    assert "cascade-fused" not in bs.plan()
    g = torch.Generator(device="cuda"); g.manual_seed(31)
    sizes = [B, B, B, 5000, B]
    xs = [torch.rand((S, n, C), dtype=torch.float64, device="cuda", generator=g) - 0.5 for n in sizes]
    L = amd.load_library()
    yf, names = [], []
    for x in xs:
        L.dspamd_profile_enable(1)
        yf.append(bf.run(x).clone())
        names.append({ln.split()[0] for ln in L.dspamd_profile_collect().decode().splitlines()})
        L.dspamd_profile_enable(0)
    assert [("fused_col_fwd" in n) for n in names] == [False, True, True, False, True], names
    ys = [bs.run(x).clone() for x in xs]
    for k, (a, b) in enumerate(zip(yf, ys)):
        assert a.shape == b.shape == (S, 2 * sizes[k] - (583 if k == 0 else 0), C), (k, a.shape, b.shape)
        assert float((a - b).pow(2).mean().sqrt()) < 1e-12, k
    tails = []
    while True:
        o = bf.drain(B)
        if o is None:
            break
        tails.append(o.clone())
    for s in sorted({0, S - 1, S // 2}):
        x = torch.cat([t[s] for t in xs], dim=0).cpu().numpy()
        ref = RefChain(chain, 48000, C).process(x, block=65536)
        got = torch.cat([t[s] for t in yf + tails if t.shape[1]], dim=0).cpu().numpy()
        assert ref.shape == got.shape, (ref.shape, got.shape)
        assert rms(ref - got) < 1e-11, (s, rms(ref - got))
    # reset: the stream starts over (a first call again: separate kernels), same bits
    bf.reset()
    assert torch.equal(bf.run(xs[0]), yf[0]) and torch.equal(bf.run(xs[1]), yf[1])


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
@pytest.mark.parametrize("S,C,sel", [(8, 8, (":0,1", ":2,3", ":4-7")), (120, 8, (":0,1", ":2,3", ":4-7")), (6, 4, (":2,3", ":0,1", ":0-3"))])
def test_sections_behind_pair_aligned_selectors_take_the_fused_path(amd, tmp_path, S, C, sel):
    """the crossover shape (biquad.c:296-305: an effect acts on its selected channels): sections and gains behind selectors that keep the two channels
    of every pair alike -- one section table per pair in the fused kernels (round 5: a wave of the first pass is 64 rows of one pair, its table
    is still read with scalar loads; the prepass reads per-lane tables), pass-through sections where a pair is not selected, gains per pair.
    Hop, hop, a call off the grid, hop: against the separate kernels and the real reference."""
    import torch
    f = os.path.join(str(tmp_path), "h.raw")
    np.asarray(make_filter(16384, seed=41), dtype="<f8").tofile(f)
    chain = (f"lowpass 6k 0.707 {sel[0]} eq 400 2.0 1.5 gain -2 lowshelf 150 0.8 2 {sel[1]} highpass 300 0.707 gain 1.5 {sel[2]} eq 3000 1.0 2 : "
             f"highshelf 8k 0.7 -3 gain -1 fir_p -t pcm -e double -c 1 {f}")
    B = 245760
    bf, bs = build(amd, chain, C, S, B, True), build(amd, chain, C, S, B, False)
    assert "cascade-fused(" in bf.plan() and "sections per pair" in bf.plan(), bf.plan()
    assert "cascade-fused" not in bs.plan()
    g = torch.Generator(device="cuda"); g.manual_seed(51)
    sizes = [B, B, 5000, B]
    xs = [torch.rand((S, n, C), dtype=torch.float64, device="cuda", generator=g) - 0.5 for n in sizes]
    L = amd.load_library()
    yf, names = [], []
    for x in xs:
        L.dspamd_profile_enable(1)
        yf.append(bf.run(x).clone())
        names.append({ln.split()[0] for ln in L.dspamd_profile_collect().decode().splitlines()})
        L.dspamd_profile_enable(0)
    assert [("fused_col_fwd" in n) for n in names] == [True, True, False, True], names
    assert "fused_prepass" in names[0] and "fused_prepass_mm" not in names[0], names[0]        # (one G per product: the recurrence form serves per-pair tables)
    ys = [bs.run(x).clone() for x in xs]
    for k, (a, b) in enumerate(zip(yf, ys)):
        assert a.shape == b.shape == (S, sizes[k], C)
        assert float((a - b).pow(2).mean().sqrt()) < 1e-12, k
    for s in sorted({0, S - 1, S // 2}):
        x = torch.cat([t[s] for t in xs], dim=0).cpu().numpy()
        ref = RefChain(chain, 48000, C).run(x)
        got = torch.cat([t[s] for t in yf], dim=0).cpu().numpy()
        assert ref.shape == got.shape
        assert rms(ref - got) < 1e-12, (s, rms(ref - got))
    bf.reset()
    assert torch.equal(bf.run(xs[0]), yf[0])


def test_chains_the_fused_kernels_leave_alone(amd, tmp_path):
    """per-channel sections, an `add` among the ops, a selector, a latency: the plan keeps the separate kernels"""
    f = os.path.join(str(tmp_path), "h.raw")
    np.asarray(make_filter(16384), dtype="<f8").tofile(f)
    g = os.path.join(str(tmp_path), "h64.raw")
    np.asarray(make_filter(65536), dtype="<f8").tofile(g)
    b = build(amd, f"lowpass 1k 0.707 fir_p -t pcm -e double -c 1 {g}", 8, 4, 196608, True)      # 64 of the 256 rows are history: the separate kernels are faster
    assert "cascade-fused" not in b.plan(), b.plan()
    # one filter per channel (fir_p.c:483-495): a pair then carries one channel, the fused first pass works on pairs of a shared filter
    g8 = os.path.join(str(tmp_path), "h8.raw")
    np.asarray(np.stack([make_filter(16384, seed=20 + c) for c in range(8)], axis=1), dtype="<f8").tofile(g8)
    b = build(amd, f"lowpass 1k 0.707 fir_p -t pcm -e double -c 8 {g8}", 8, 4, 245760, True)
    assert "per-channel-filters" in b.plan() and "cascade-fused" not in b.plan(), b.plan()
    for chain in (f"lowpass 1k 0.707 :0 eq 400 2.0 1.5 : fir_p -t pcm -e double -c 1 {f}",
                  f"lowpass 1k 0.707 add 0.001 fir_p -t pcm -e double -c 1 {f}",
                  f"gain -3 fir_p -t pcm -e double -c 1 {f}",
                  f"lowpass 1k 0.707 fir -t pcm -e double -c 1 {f}"):
        b = build(amd, chain, 8, 4, 245760, True)
        assert "cascade-fused" not in b.plan(), b.plan()


@pytest.mark.parametrize("in_fmt,out_fmt,prec", [("s16", "s16", 16), ("s32", "s24", 24), ("float", "float", 0), ("s24", "double", 0)])
def test_wire_formats_into_the_fused_first_pass(amd, tmp_path, in_fmt, out_fmt, prec):
    """the headline chain's shape in small from wire format to wire format: the matrix-core prepass and the fused first pass read the
    samples themselves (read_buf_<fmt> in their loads), K3 applies the sink -- every byte, clip count and peak equal to the same
    conversions as passes of their own around the same (fused) fp64 step; hop, hop, a call off the grid, hop"""
    import torch
    import test_gpu_wire as tw
    mods = (amd, amd.load_library(), torch)
    f = os.path.join(str(tmp_path), "h.raw")
    np.asarray(make_filter(16384, seed=5), dtype="<f8").tofile(f)
    chain = "gain 3 " + " ".join(SECTIONS[:10]) + f" fir_p -t pcm -e double -c 1 {f}"
    S, C, B = 8, 8, 245760
    blocks = [B, B, 5000, B]
    x = tw.wire_input(torch, in_fmt, S, sum(blocks), C, 77)
    got, gstats, bits = tw.fused(mods, chain, 48000, C, S, x, blocks, in_fmt, out_fmt, prec, pad=68)
    want, wstats, plan = tw.separate_passes(mods, chain, 48000, C, S, x, blocks, in_fmt, out_fmt, prec)
    assert "cascade-fused" in plan, plan
    assert tw.same(got, want)
    assert np.array_equal(gstats, wstats)
    if in_fmt != "double":
        tw.check_bits(bits, [3, 3, 3, 3])          # both ends converted inside kernels, in all four calls


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
def test_fused_path_with_poles_next_to_the_unit_circle(amd, tmp_path):
    """sections whose states decay over hundreds of thousands of frames (5 Hz high-pass, 12 Hz resonance of Q 12, a 30 Hz low-pass): the
    end states of the rows come out of a 1024-term product with a slowly decaying table and a scan with powers of a matrix close to
    the identity -- held to the reference's plain recurrence over three hops, at 3e-12 of the input level"""
    import torch
    f = os.path.join(str(tmp_path), "h.raw")
    np.asarray(make_filter(16384, seed=9), dtype="<f8").tofile(f)
    chain = f"highpass 5 0.707 eq 12 12.0 9 lowpass 30 0.5 gain -20 highshelf 9k 0.7 4 fir_p -t pcm -e double -c 1 {f}"
    S, C, B = 8, 8, 245760
    bf, bs = build(amd, chain, C, S, B, True), build(amd, chain, C, S, B, False)
    assert "cascade-fused" in bf.plan() and "cascade-fused" not in bs.plan()
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    xs = [torch.rand((S, B, C), dtype=torch.float64, device="cuda", generator=g) - 0.5 + 0.25 for _ in range(3)]     # (a DC offset: the high-pass has something to forget)
    yf = [bf.run(x).clone() for x in xs]
    ys = [bs.run(x).clone() for x in xs]
    for s in (0, 5):
        x = torch.cat([t[s] for t in xs], dim=0).cpu().numpy()
        ref = RefChain(chain, 48000, C).run(x)
        for name, ys_ in (("fused", yf), ("separate", ys)):
            got = torch.cat([t[s] for t in ys_], dim=0).cpu().numpy()
            # the chain's output lies 2000x below its inner states (a 30 Hz low-pass behind a resonance), so the rounding of ANY
            # evaluation order shows at 1e-9 of the output: measured 1.0e-9 (fused) and 6e-10 (separate kernels) -- the tolerance
            # is stated against the level of the input, where both are 3e-13 .. 6e-13
            e = rms(ref - got) / rms(x)
            assert e < 3e-12, (name, s, e, rms(ref), rms(x))


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
@pytest.mark.parametrize("taps,B,S,C", [(16384, 245760, 8, 8), (32768, 229376, 3, 8), (16384, 245760, 130, 4), (11111, 245760, 4, 8)])      # (the last: a filter shorter than 16 rows takes 16 whole rows of history)
def test_fir_p_first_in_the_chain_takes_the_two_pair_first_pass(amd, tmp_path, taps, B, S, C):
    """no cascade in front (BASELINE config 3's shape): calls of one whole hop go through the fused first pass with a pass-through section and zero
    states instead of K1's slab-direct form -- the same samples into the same transform.  Against the separate kernels and the real reference,
    across a call off the grid (K1 on the rings the fused pass filed) and back."""
    import torch
    f = os.path.join(str(tmp_path), "h.raw")
    np.asarray(make_filter(taps, seed=taps + 1), dtype="<f8").tofile(f)
    chain = f"fir_p -t pcm -e double -c 1 {f}"
    bf, bs = build(amd, chain, C, S, B, True), build(amd, chain, C, S, B, False)
    assert "two pairs per workgroup" in bf.plan(), bf.plan()
    assert "two pairs per workgroup" not in bs.plan(), bs.plan()
    g = torch.Generator(device="cuda"); g.manual_seed(77)
    sizes = [B, B, 5000, B, 8 * 100, B]          # (the last hop starts at a multiple of 8 again: fused)
    xs = [torch.rand((S, n, C), dtype=torch.float64, device="cuda", generator=g) - 0.5 for n in sizes]
    yf = [bf.run(x).clone() for x in xs]
    ys = [bs.run(x).clone() for x in xs]
    for a, b_ in zip(yf, ys):
        assert float((a - b_).abs().max()) < 1e-13
    for s in sorted({0, S // 2, S - 1}):
        x = torch.cat([t[s] for t in xs], dim=0).cpu().numpy()
        ref = RefChain(chain, 48000, C).run(x)
        got = torch.cat([t[s] for t in yf], dim=0).cpu().numpy()
        assert rms(ref - got) < 1e-12, (s, rms(ref - got))
    bf.reset()
    assert torch.equal(bf.run(xs[0]), yf[0])
