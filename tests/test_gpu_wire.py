"""Wire format to wire format in ONE pipeline (fused): read_buf_<fmt> in the first kernel's loads,
dither / clip() / write_buf_<fmt> (dsp.c:685-699) in the last kernel's stores, through dspamd_batch_run_wire.

The bar is bit-exactness against the SAME conversions done as passes of their own (dspamd_pcm_read -> dspamd_batch_run ->
dspamd_pcm_write, which tests/test_pcm.py pins bit for bit against the reference's functions and the stock CLI's bytes): the
fused kernels do the same IEEE operations on the same values, so every byte, every clip count and every peak must agree --
for every format, with and without dither, across calls of ragged sizes (the dither sequences run on through the stream) and
through the drain.  One test compares with the bytes the reference CLI itself writes for an s16 -> s16 file run."""
import os
import subprocess

import numpy as np
import pytest

from oracle_api import REF_DIR

pytestmark = pytest.mark.gpu

DSP_REF = os.path.join(REF_DIR, "dsp_ref")
NP_DT = {"u8": np.uint8, "s8": np.int8, "s16": np.int16, "s24": np.int32, "s32": np.int32, "s24_3": np.uint8, "float": np.float32, "double": np.float64}
EQ10 = " ".join(f"eq {f} 1.2 {g}" for f, g in zip((60, 120, 250, 500, 1000, 2000, 4000, 8000, 12000, 16000), (1.5, -2, 1, -1, 2, -1.5, 1, -2, 1.5, -1)))


@pytest.fixture(scope="module")
def gpu():
    import torch
    import dsp_amd
    L = dsp_amd.load_library()
    assert L.dspamd_device_count() >= 1
    return dsp_amd, L, torch


def wire_input(torch, fmt, S, F, Cn, seed):
    """[S, F, C] samples of the wire format (device), loud enough to clip after a boost"""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.uniform(-0.9, 0.9, size=(S, F, Cn))
    if fmt == "double":
        return torch.from_numpy(x).cuda()
    if fmt == "float":
        return torch.from_numpy(x.astype(np.float32)).cuda()
    if fmt == "s16":
        return torch.from_numpy(np.round(x * 32767).astype(np.int16)).cuda()
    if fmt == "s24":
        return torch.from_numpy((np.round(x * 8388607).astype(np.int32)) & 0xffffff).to(torch.int32).cuda()   # junk-free upper byte not required: sign-extension is the kernel's job
    if fmt == "s32":
        return torch.from_numpy(np.round(x * 2147483647).astype(np.int64).astype(np.int32)).cuda()
    if fmt == "u8":
        return torch.from_numpy(np.round(x * 127 + 128).astype(np.uint8)).cuda()
    if fmt == "s8":
        return torch.from_numpy(np.round(x * 127).astype(np.int8)).cuda()
    if fmt == "s24_3":
        v = np.round(x * 8388607).astype(np.int32)
        b = np.stack([(v & 0xff), ((v >> 8) & 0xff), ((v >> 16) & 0xff)], axis=-1).astype(np.uint8)
        return torch.from_numpy(b.reshape(S, F, Cn * 3)).cuda()
    raise ValueError(fmt)


def separate_passes(mods, chain, fs, Cn, S, x, blocks, in_fmt, out_fmt, prec, filt_dir=None):
    """the conversions as passes of their own around dspamd_batch_run: (bytes [S, M, ...], stats [S, 2])"""
    dsp_amd, L, torch = mods
    from dsp_amd.chain import PCM_FORMATS, WIRE_DTYPES
    b = dsp_amd.BatchChain(chain, fs, Cn, S, max(blocks), directory=filt_dir)
    stats = torch.zeros((S, 2), dtype=torch.float64, device="cuda")
    outs, pos, written = [], 0, 0
    mult_in = WIRE_DTYPES[in_fmt][1]

    def sink(y):
        nonlocal written
        f = y.shape[1]
        if f == 0:
            return
        yc = y.contiguous()
        dt, mult = WIRE_DTYPES[out_fmt]
        o = torch.empty((S, f, b.ochannels * mult), dtype=getattr(torch, dt), device="cuda")
        assert L.dspamd_pcm_write(PCM_FORMATS[out_fmt], yc.data_ptr(), f, o.data_ptr(), S, f, b.ochannels, prec, written, stats.data_ptr(), None) == 0
        written += f
        outs.append(o)

    for n in blocks:
        seg = x[:, pos:pos + n, :].contiguous()
        pos += n
        d = torch.empty((S, n, Cn), dtype=torch.float64, device="cuda")
        assert L.dspamd_pcm_read(PCM_FORMATS[in_fmt], seg.data_ptr(), d.data_ptr(), S * n * Cn, None) == 0
        assert seg.shape[2] == Cn * mult_in
        sink(b.run(d))
    while True:
        y = b.drain(max(blocks))
        if y is None:
            break
        sink(y)
    torch.cuda.synchronize()
    plan = b.plan()
    b.close()
    return torch.cat(outs, dim=1).cpu().numpy(), stats.cpu().numpy(), plan


def fused(mods, chain, fs, Cn, S, x, blocks, in_fmt, out_fmt, prec, filt_dir=None, pad=0):
    dsp_amd, L, torch = mods
    b = dsp_amd.BatchChain(chain, fs, Cn, S, max(blocks), directory=filt_dir)
    stats = torch.zeros((S, 2), dtype=torch.float64, device="cuda")
    outs, pos, bits = [], 0, []
    for n in blocks:
        seg = x[:, pos:pos + n, :]
        pos += n
        if pad:
            buf = torch.zeros((S, n + pad, x.shape[2]), dtype=x.dtype, device="cuda")
            buf[:, :n, :] = seg
            seg = buf[:, :n, :]
        else:
            seg = seg.contiguous()
        outs.append(b.run_wire(seg, in_fmt, out_fmt, prec, stats).clone())
        bits.append(b.wire_fused())
    while True:
        y = b.drain_wire(max(blocks), out_fmt, prec, stats)
        if y is None:
            break
        outs.append(y.clone())
    torch.cuda.synchronize()
    b.close()
    outs = [o for o in outs if o.shape[1]]
    return torch.cat(outs, dim=1).cpu().numpy(), stats.cpu().numpy(), bits


def check_bits(bits, want, note=None):
    """which ends were fused is asserted on the default plan only: the fallback suite's DSP_AMD_* switches change the kernels"""
    if any(k.startswith("DSP_AMD_") for k in os.environ):
        return
    want = want if isinstance(want, (list, tuple)) else [want] * len(bits)
    assert list(bits) == list(want), (bits, want, note)


def same(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def write_filter(tmp_path, taps, seed=3):
    rng = np.random.Generator(np.random.PCG64(seed))
    h = rng.standard_normal(taps) * np.exp(-np.arange(taps) / (taps / 6.0))
    h = h / np.sqrt(np.sum(h * h)) / 2.0
    path = os.path.join(str(tmp_path), f"h{taps}.raw")
    np.asarray(h, dtype="<f8").tofile(path)
    return path, h


# the headline plan's shape in small: identical sections on every channel, >= 1024 channels (cascade_rows<4>), a zero-latency
# convolution behind it (K3 writes the output)
@pytest.mark.parametrize("in_fmt,out_fmt,prec", [("s16", "s16", 16), ("s32", "s24", 24), ("float", "float", 0), ("s24", "s32", 0),
                                                 ("double", "s16", 0), ("s16", "double", 0), ("float", "s16", 16)])
def test_cascade_in_k3_out(gpu, tmp_path, in_fmt, out_fmt, prec):
    path, _ = write_filter(tmp_path, 3000)
    chain = f"gain 7 {EQ10} fir_p -t pcm -e double -c 1 {path}"
    S, Cn, fs = 128, 8, 48000
    blocks = [4096, 2600, 4096, 1111]
    x = wire_input(gpu[2], in_fmt, S, sum(blocks), Cn, 1)
    want, wstats, plan = separate_passes(gpu, chain, fs, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    got, gstats, bits = fused(gpu, chain, fs, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    assert "cascade" in plan and "conv[" in plan
    assert same(got, want)
    assert np.array_equal(gstats.view(np.uint64), wstats.view(np.uint64))
    assert wstats[:, 1].max() > 1.0 and wstats.view(np.uint64)[:, 0].min() > 0     # the boost clips: the statistics are exercised
    # every long call: input converted by cascade_rows (unless it already is fp64), sink applied by K3
    want_bits = (0 if in_fmt == "double" else 1) | 2
    check_bits(bits, want_bits)


# a chain that is ONE cascade stage: both conversions in the same kernel; padded input slabs; the drain
@pytest.mark.parametrize("S,Cn,bits_want", [(128, 8, 3), (64, 8, 3), (3, 2, 3), (40, 4, 3), (5, 3, 0)])   # cascade_rows<4>, <2>, and <1> (one channel per wave: 8- / 4- / 2-byte elements); an odd channel count runs kernels that do not speak the formats
@pytest.mark.parametrize("in_fmt,out_fmt,prec", [("s16", "s16", 16), ("s32", "float", 0), ("float", "s24", 20)])
def test_cascade_both_ends(gpu, S, Cn, bits_want, in_fmt, out_fmt, prec):
    chain = f"gain 8 {EQ10}"
    blocks = [5000, 2048, 3333]
    x = wire_input(gpu[2], in_fmt, S, sum(blocks), Cn, 2)
    want, wstats, _ = separate_passes(gpu, chain, 44100, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    got, gstats, bits = fused(gpu, chain, 44100, Cn, S, x, blocks, in_fmt, out_fmt, prec, pad=17 * 4)
    assert same(got, want)
    assert np.array_equal(gstats.view(np.uint64), wstats.view(np.uint64))
    check_bits(bits, bits_want)


# formats and plans the kernels do not speak themselves: the stand-alone passes run inside run_wire -- same bytes, bits say so
@pytest.mark.parametrize("chain,S,Cn,in_fmt,out_fmt,prec,bits_want", [
    (f"gain 8 {EQ10}", 64, 8, "s24_3", "u8", 8, 0),             # unaligned formats
    (f"gain 8 {EQ10}", 64, 8, "s8", "s24_3", 24, 0),
    (f"gain 8 {EQ10}", 64, 8, "s16", "s24_3", 0, 1),            # input fusable, output not
    (f"gain 8 {EQ10}", 64, 8, "u8", "s16", 16, 2),              # and the other way round
    (f"gain 8 {EQ10}", 16, 2, "s16", "s16", 16, (3, 0, 3)),     # few channels: cascade_rows<1> speaks them since round 3 (calls of at least one 2048-frame tile)
    (f"gain 8 {EQ10}", 16, 2, "s16", "s24_3", 24, (1, 0, 1)),
    ("gain 3 resample 44.1k", 16, 2, "s16", "s16", 16, 2),      # rate changer last: the GEMM resampler applies the sink sample by sample (round 3); its drain goes through the stand-alone sink
    ("resample 44.1k", 4, 10, "s16", "s24_3", 24, 2),           # more than 8 channels: the dot-product resampler; element-wise stores take any format
    ("resample 32k", 6, 3, "float", "u8", 8, 2),
    ("", 16, 2, "s16", "float", 0, 0),                          # no effects at all: conversion only
])
def test_unfused_paths_inside_run_wire(gpu, chain, S, Cn, in_fmt, out_fmt, prec, bits_want):
    blocks = [4096, 1100, 4096]
    x = wire_input(gpu[2], in_fmt, S, sum(blocks), Cn, 3)
    want, wstats, _ = separate_passes(gpu, chain, 48000, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    got, gstats, bits = fused(gpu, chain, 48000, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    assert same(got, want)
    assert np.array_equal(gstats.view(np.uint64), wstats.view(np.uint64))
    check_bits(bits, bits_want)


# the other first / last kernels: element-wise stages (remix, the alignment delay) speak every format; a convolver at the start
# reads the wire format in K1 (all channels, aligned pairs) or in its de-interleaving pass; K3 at the end applies the sink, also
# on single channels of a pair; `fir` ends in the alignment stage that discards its latency
@pytest.mark.parametrize("chain,S,Cn,in_fmt,out_fmt,prec,bits_want", [
    ("remix 1 0", 16, 2, "s24_3", "u8", 8, 3),
    ("remix 0,1 0 1 gain 2", 16, 2, "s16", "s16", 16, 1),       # remix first, a 48-channel cascade last
    ("delay 2m", 16, 2, "s16", "s24_3", 24, 3),                 # one alignment stage
    ("fir_p -t pcm -e double -c 1 {F}", 128, 8, "s16", "s16", 16, 3),     # K1 direct in, K3 out
    ("fir_p -t pcm -e double -c 1 {F}", 128, 8, "s24_3", "s16", 16, 2),   # K1 cannot: the read pass runs; K3 out
    ("fir_p -t pcm -e double -c 1 {F}", 24, 3, "s8", "s32", 0, 3),        # odd channels: the de-interleaving pass in; K3 out (single channel of a pair)
    (f"{EQ10} fir -t pcm -e double -c 1 {{F}}", 24, 3, "s16", "s16", 16, 2),   # cascade (few channels) ... fir, alignment stage out
    ("gain 6 fir_p -t pcm -e double -c 1 {F}", 24, 5, "float", "float", 0, 2),
    ("resample 96k", 32, 2, "s16", "s16", 16, 3),               # 2x upsampler first and last: K1 + its history pass in; one pair per stream: the general K3 applies the sink sample by sample (round 3)
    ("resample 24k", 16, 8, "s16", "s16", 16, 3),               # 2:1 decimator, four pairs per stream: every second frame of the general K3 is a sample
    ("resample 144k", 8, 2, "float", "s24", 24, 3),             # three phases
    ("resample 192k", 6, 4, "s32", "float", 0, 3),              # four phases, two pairs per stream
    ("resample 16k", 5, 3, "s16", "s32", 0, 3),                 # 3:1 decimator, odd channel count
    ("resample 96k", 128, 8, "s16", "s16", 16, 3),              # ... four pairs per stream: the two-phase K3 applies the sink (frames 2q, 2q + 1 per thread)
    ("resample 96k", 128, 8, "float", "s24", 24, 3),
    (f"{EQ10} fir_p -t pcm -e double -c 1 {{F}} resample 96k", 128, 8, "s16", "s16", 16, (3, 3, 2)),   # BASELINE config 4's chain: cascade in (calls of at least one 512-frame tile), merged fir_p + 2x upsampler out; its drain tail goes through the stand-alone sink
    ("resample 44.1k", 32, 2, "s16", "s16", 16, 2),             # the general resampler: input through the read pass, the sink in the GEMM kernel's stores
    ("fir -t pcm -e double -c 1 {F}", 64, 8, "s16", "s16", 16, 3),        # `fir` on every channel: K3 drops the latency frames itself (no alignment pass) and applies the sink
    ("zita_convolver -t pcm -e double -c 1 {F}", 16, 2, "float", "float", 0, 3),   # float32 in -> float32-spectrum stage -> float32 out: the zita contract from wire to wire
    ("zita_convolver -t pcm -e double -c 1 {F}", 16, 2, "s16", "s24", 24, 3),
])
def test_other_first_and_last_kernels(gpu, tmp_path, chain, S, Cn, in_fmt, out_fmt, prec, bits_want):
    path, _ = write_filter(tmp_path, 700)
    chain = chain.replace("{F}", path)
    blocks = [3000, 3000, 500]
    x = wire_input(gpu[2], in_fmt, S, sum(blocks), Cn, 4)
    want, wstats, plan = separate_passes(gpu, chain, 48000, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    got, gstats, bits = fused(gpu, chain, 48000, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    assert same(got, want), plan
    assert np.array_equal(gstats.view(np.uint64), wstats.view(np.uint64))
    check_bits(bits, bits_want, plan)


# the one-trip convolver in the middle of a long call: whole windows read from the wire-format slab by straight-line loads (the blocks above are shorter
# than a hop: one partial window per call), on both of its window sizes
@pytest.mark.parametrize("taps,blocks,in_fmt,out_fmt,prec,n_window", [
    (700, [20000, 17000, 500], "s16", "s16", 16, 8192),
    (700, [20000, 17000, 500], "float", "s24", 24, 8192),
    (3000, [60000, 30000, 500], "s16", "s16", 16, 16384),
    (3000, [60000, 30000, 500], "s32", "float", 0, 16384),
])
def test_one_trip_whole_windows_speak_the_formats(gpu, tmp_path, taps, blocks, in_fmt, out_fmt, prec, n_window):
    path, _ = write_filter(tmp_path, taps)
    chain = f"fir_p -t pcm -e double -c 1 {path}"
    S, Cn = 12, 4
    x = wire_input(gpu[2], in_fmt, S, sum(blocks), Cn, 4)
    want, wstats, plan = separate_passes(gpu, chain, 48000, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    assert "one-trip" in plan and f"N={n_window}=" in plan, plan
    got, gstats, bits = fused(gpu, chain, 48000, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    assert same(got, want), plan
    assert np.array_equal(gstats.view(np.uint64), wstats.view(np.uint64))
    check_bits(bits, 3, plan)


# ragged calls: whole tiles + a remainder (the generic kernel speaks the formats behind cascade_rows' tiles), calls shorter than a
# tile (stand-alone passes inside the call), a single frame -- the dither sequences run on through all of them
@pytest.mark.parametrize("chain_tail", ["", " fir_p -t pcm -e double -c 1 {F}"])
def test_ragged_calls_keep_the_dither_sequences(gpu, tmp_path, chain_tail):
    path, _ = write_filter(tmp_path, 900)
    chain = f"gain 8 {EQ10}" + chain_tail.replace("{F}", path)
    S, Cn = 128, 8
    blocks = [4096, 100, 1, 4096, 700, 513, 511]
    x = wire_input(gpu[2], "s16", S, sum(blocks), Cn, 7)
    want, wstats, _ = separate_passes(gpu, chain, 48000, Cn, S, x, blocks, "s16", "s16", 16)
    got, gstats, bits = fused(gpu, chain, 48000, Cn, S, x, blocks, "s16", "s16", 16)
    assert same(got, want)
    assert np.array_equal(gstats.view(np.uint64), wstats.view(np.uint64))
    out_bit = 2                                                   # cascade_rows (chain of sections) or K3 (convolver last) ...
    check_bits(bits, [3, 0 if not chain_tail else out_bit, 0 if not chain_tail else out_bit, 3, 3, 3, 0 if not chain_tail else out_bit])


# reset: the dither sequences start again
def test_reset_restarts_the_dither_sequences(gpu):
    dsp_amd, L, torch = gpu
    chain = f"gain 2 {EQ10}"
    S, Cn, n = 128, 8, 4096
    x = wire_input(torch, "s16", S, n, Cn, 5)
    b = dsp_amd.BatchChain(chain, 48000, Cn, S, n)
    a = b.run_wire(x, "s16", "s16", 16).clone()
    c = b.run_wire(x, "s16", "s16", 16).clone()
    b.reset()
    d = b.run_wire(x, "s16", "s16", 16).clone()
    assert torch.equal(a, d) and not torch.equal(a, c)


# the switch that forces the stand-alone passes exists for the fallback suite; it must say so in the bits
def test_no_fusion_switch_is_honoured_in_a_fresh_process(tmp_path):
    code = "\n".join([
        "import torch, dsp_amd",
        "b = dsp_amd.BatchChain('gain 2 eq 1k 1 3 eq 2k 1 -3', 48000, 8, 128, 4096)",
        "x = torch.zeros((128, 4096, 8), dtype=torch.int16, device='cuda')",
        "b.run_wire(x, 's16', 's16', 16)",
        "print('BITS', b.wire_fused())",
    ])
    env = dict(os.environ, DSP_AMD_NO_WIRE_FUSION="1")
    r = subprocess.run(["python", "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert "BITS 0" in r.stdout, r.stdout[-2000:]


# the reference CLI, file to file: s16 in, ten sections, s16 out with dither.  The chain itself agrees with the reference to
# ~1e-13 (not bit for bit: the sections run as a scan here), so a sample can land on the other side of a rounding boundary
# once in ~1e8: the bytes are expected to be identical, and at most a handful of +-1 LSB differences are tolerated
@pytest.mark.skipif(not os.path.exists(DSP_REF), reason="oracle/_ref/dsp_ref not built")
def test_file_to_file_against_the_reference_cli(gpu, tmp_path):
    dsp_amd, L, torch = gpu
    Cn, F = 8, 60000
    rng = np.random.Generator(np.random.PCG64(6))
    pcm = np.round(rng.uniform(-0.5, 0.5, size=(F, Cn)) * 32767).astype("<i2")
    xin = os.path.join(str(tmp_path), "in.raw"); pcm.tofile(xin)
    out = os.path.join(str(tmp_path), "out.raw")
    eff = f"gain 5 {EQ10}".split()
    cmd = [DSP_REF, "-q", "-d", "-t", "pcm", "-e", "s16", "-r", "48k", "-c", str(Cn), xin, "-o", "-t", "pcm", "-e", "s16", out] + eff
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    want = np.fromfile(out, dtype="<i2").reshape(-1, Cn)
    # 128 streams so that cascade_rows<4> takes it; stream 0 carries the file
    S = 128
    x = torch.zeros((S, F, Cn), dtype=torch.int16, device="cuda")
    x[0] = torch.from_numpy(pcm.astype(np.int16)).cuda()
    b = dsp_amd.BatchChain(" ".join(eff), 48000, Cn, S, 8192)
    y = b.process_wire(x, 8192, "s16", "s16", 16)
    got = y[0].cpu().numpy()
    assert got.shape == want.shape
    d = got.astype(np.int32) - want.astype(np.int32)
    assert np.abs(d).max() <= 1 and np.count_nonzero(d) <= 4, (np.abs(d).max(), np.count_nonzero(d))


@pytest.mark.parametrize("in_fmt,out_fmt,prec", [("s16", "s16", 16), ("double", "float", 0), ("s32", "s24", 24), ("float", "double", 0)])
def test_persistent_k3_speaks_the_formats(gpu, tmp_path, monkeypatch, in_fmt, out_fmt, prec):
    # a shape that reaches the persistent K3 (streams of four pairs, N = 2^18: 8 streams x 128 column blocks = 1024 tiles):
    # the plain call and the call with the sink must run the SAME kernel instance -- two instances of one FFT source are not
    # guaranteed the same bits (docs/history.md section 4.6) -- so every byte and both statistics agree with the stand-alone passes
    monkeypatch.setenv("DSP_AMD_CONV_LOG2N", "18")
    path, _ = write_filter(tmp_path, 3000)
    chain = f"fir_p -t pcm -e double -c 1 {path}"
    S, Cn, fs = 8, 8, 48000
    blocks = (5000, 4096, 777)
    x = wire_input(gpu[2], in_fmt, S, sum(blocks), Cn, 31)
    a, sa, plan = separate_passes(gpu, chain, fs, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    assert "N=262144" in plan
    f, sf, bits = fused(gpu, chain, fs, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    assert same(a, f)
    assert np.array_equal(sa, sf)
    check_bits(bits[:3], [3 if in_fmt in ("s16", "s32", "s24", "float") else 2] * 3)


@pytest.mark.parametrize("in_fmt,out_fmt,prec", [("s16", "s16", 16), ("float", "s24", 24)])
def test_mid_size_calls_speak_the_formats(gpu, tmp_path, in_fmt, out_fmt, prec):
    # calls of 8192 frames on a 20000-tap filter: the delay-line form of the convolver (a child stage on the same rings) takes the
    # sink in its K3 like the plain form; the last, short call leaves the grid and goes through the plain form
    path, _ = write_filter(tmp_path, 20000)
    chain = f"fir_p -t pcm -e double -c 1 {path}"
    S, Cn, fs = 16, 8, 48000
    blocks = (8192, 8192, 8192, 8192, 900)
    x = wire_input(gpu[2], in_fmt, S, sum(blocks), Cn, 77)
    a, sa, plan = separate_passes(gpu, chain, fs, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    assert "mid-size-calls" in plan or os.environ.get("DSP_AMD_CONV_UPC") == "0"
    f, sf, bits = fused(gpu, chain, fs, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    assert same(a, f)
    assert np.array_equal(sa, sf)
    check_bits(bits[:4], [3] * 4)          # (input converted by the de-interleaving pass that fills the rings; K3 applies the sink)


@pytest.mark.parametrize("S,Cn,head,taps,blocks", [
    (16, 8, "", 40000, [2048] * 9 + [300]),                                   # head 4 x 2048 + delay-line tail
    (6, 3, "gain -1 ", 20000, [1024] * 9 + [300]),                            # odd channel count: a single channel in the last pair
    (8, 2, "", 9000, [512] * 9 + [300]),                                      # 1024-point rows: four pairs per workgroup, streams differ inside a workgroup
    (128, 8, EQ10 + " ", 70000, [4096] * 3 + [2048] * 3 + [300]),             # two sub-blocks per launch and one; cascade_rows in front (reads the wire format)
])
@pytest.mark.parametrize("in_fmt,out_fmt,prec", [("s16", "s16", 16), ("float", "s24", 24), ("s32", "double", 0)])
def test_small_calls_speak_the_formats(gpu, tmp_path, S, Cn, head, taps, blocks, in_fmt, out_fmt, prec):
    # calls much shorter than the filter: the delay-line kernel (conv_fdl) is the last kernel of the call and applies the sink itself --
    # pairs of adjacent channels as 16- / 8- / 4-byte stores, the single channel of an odd count sample by sample; sub-blocks of one
    # launch and the launches of one call continue the dither sequence; the last, ragged call goes through the plain form's K3
    path, _ = write_filter(tmp_path, taps)
    chain = f"{head}fir_p -t pcm -e double -c 1 {path}"
    x = wire_input(gpu[2], in_fmt, S, sum(blocks), Cn, 91)
    a, sa, plan = separate_passes(gpu, chain, 48000, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    assert "small-calls" in plan or any(k.startswith("DSP_AMD_") for k in os.environ), plan
    f, sf, bits = fused(gpu, chain, 48000, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    assert same(a, f)
    assert np.array_equal(sa, sf)
    check_bits([b & 2 for b in bits], [2] * len(bits))          # the sink is applied by the stage's own last kernel in every call, on the grid or off it


@pytest.mark.parametrize("S,Cn", [(32, 8), (12, 4)])
@pytest.mark.parametrize("in_fmt,out_fmt,prec", [("s16", "s16", 16), ("float", "s32", 0), ("double", "s24", 24)])
def test_strong_scaling_cascade_speaks_the_formats(gpu, S, Cn, in_fmt, out_fmt, prec):
    # fewer than 512 channels: cascade_rows<1>, one channel per workgroup of time-skewed waves -- 8- / 4- / 2-byte elements a frame apart.
    # Long calls: every wave walks several tiles (its dither generators jump P - 1 tiles between them), point-to-point ordering on
    chain = f"gain 8 {EQ10}"
    blocks = [43000, 2048 * 9 + 77]
    x = wire_input(gpu[2], in_fmt, S, sum(blocks), Cn, 5)
    want, wstats, plan = separate_passes(gpu, chain, 48000, Cn, S, x, blocks, in_fmt, out_fmt, prec)
    got, gstats, bits = fused(gpu, chain, 48000, Cn, S, x, blocks, in_fmt, out_fmt, prec, pad=8)
    assert same(got, want)
    assert np.array_equal(gstats.view(np.uint64), wstats.view(np.uint64))
    assert wstats[:, 1].max() > 1.0 and wstats.view(np.uint64)[:, 0].min() > 0     # the boost clips: the statistics are exercised
    check_bits(bits, (0 if in_fmt == "double" else 1) | 2)
