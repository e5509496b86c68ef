"""Wire formats either side of the chain: read_buf_<fmt> / write_buf_<fmt>
(sampleconv.c, BIT_PERFECT macros), clip() and the TPDF dither of the output stage (dsp.c:673-699, util.h:127-178).
Everything here is bit-exact.
  CPU: the oracle restatement against the real reference's own conversion functions (in-process, oracle/_ref/libdspref.so)
       and against the bytes the stock reference CLI writes (oracle/_ref/dsp_ref, dither forced with -d).
  GPU: the device kernels (through the C ABI) against the oracle, and against the CLI's bytes."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle_api import ORACLE_SO, REF_DIR, RefChain

FMTS = {"u8": (0, np.uint8), "s8": (1, np.int8), "s16": (2, np.int16), "s24": (3, np.int32), "s32": (4, np.int32),
        "s24_3": (5, np.uint8), "float": (6, np.float32), "double": (7, np.float64)}
BYTES = {"u8": 1, "s8": 1, "s16": 2, "s24": 4, "s32": 4, "s24_3": 3, "float": 4, "double": 8}
DSP_REF = os.path.join(REF_DIR, "dsp_ref")


def orc():
    L = C.CDLL(ORACLE_SO)
    L.orc_pcm_read.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_ssize_t]
    L.orc_pcm_write.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_int, C.c_void_p, C.c_void_p]
    return L


def orc_write(x, fmt, dither_prec=0, state=None):
    L = orc()
    out = np.zeros(x.size * BYTES[fmt], dtype=np.uint8)
    st = np.array([1, 1], dtype=np.uint32) if state is None else state
    stats = np.zeros(2)
    xx = np.ascontiguousarray(x, dtype=np.float64)
    L.orc_pcm_write(FMTS[fmt][0], xx.ctypes.data, out.ctypes.data, xx.size, dither_prec, st.ctypes.data, stats.ctypes.data)
    return out, st, stats


def orc_read(raw, fmt):
    L = orc()
    n = raw.size // BYTES[fmt]
    out = np.zeros(n)
    rr = np.ascontiguousarray(raw)
    L.orc_pcm_read(FMTS[fmt][0], rr.ctypes.data, out.ctypes.data, n)
    return out


def test_signal(n, seed, amp=1.3):
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.uniform(-amp, amp, size=n)
    x[:8] = [0.0, 1.0, -1.0, 0.99999, -0.99999, 1.0 - 2.0 ** -16, 0.5 / 32768.0, 1.5 / 32768.0]   # edges, ties
    return x
test_signal.__test__ = False


@pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")
@pytest.mark.parametrize("fmt", list(FMTS))
def test_oracle_conversions_equal_the_reference_functions(fmt):
    R = RefChain.lib()
    x = np.clip(test_signal(5000, 3), -1.0, 1.0)
    want = np.zeros(x.size * BYTES[fmt], dtype=np.uint8)
    xx = x.copy()
    getattr(R, f"write_buf_{fmt}")(C.c_void_p(xx.ctypes.data), C.c_void_p(want.ctypes.data), C.c_ssize_t(x.size))
    got, _, _ = orc_write(x, fmt)
    assert np.array_equal(got, want)
    back_want = np.zeros(x.size)
    getattr(R, f"read_buf_{fmt}")(C.c_void_p(want.ctypes.data), C.c_void_p(back_want.ctypes.data), C.c_ssize_t(x.size))
    assert np.array_equal(orc_read(want, fmt), back_want)


def cli_bytes(tmp_path, x, channels, enc, dither):
    xin = os.path.join(str(tmp_path), "in.raw"); np.asarray(x, dtype="<f8").tofile(xin)
    out = os.path.join(str(tmp_path), "out.raw")
    cmd = [DSP_REF, "-q", "-d" if dither else "-D", "-t", "pcm", "-e", "double", "-r", "48k", "-c", str(channels), xin,
           "-o", "-t", "pcm", "-e", enc, out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    return np.fromfile(out, dtype=np.uint8)


@pytest.mark.skipif(not os.path.exists(DSP_REF), reason="oracle/_ref/dsp_ref not built")
@pytest.mark.parametrize("enc,prec,dither", [("s16", 16, True), ("s16", 16, False), ("s24_3", 24, True), ("u8", 8, True), ("s32", 32, False)])
def test_oracle_sink_equals_the_reference_cli(tmp_path, enc, prec, dither):
    x = test_signal(2 * 6001, 4)
    want = cli_bytes(tmp_path, x, 2, enc, dither)
    got, _, stats = orc_write(x, enc, prec if dither else 0)
    assert np.array_equal(got, want)
    assert stats[0] > 0 and stats[1] > 1.0          # the signal clips on purpose


# ------------------------------------------------------------------------------------------------ GPU

@pytest.fixture(scope="module")
def gpu():
    import torch
    import dsp_amd
    L = dsp_amd.load_library()
    assert L.dspamd_device_count() >= 1
    return L, torch


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", list(FMTS))
def test_device_read_write_equal_the_oracle(gpu, fmt):
    L, torch = gpu
    S, F, Cn = 3, 4097, 2
    x = np.stack([test_signal(F * Cn, 10 + s).reshape(F, Cn) for s in range(S)])
    dx = torch.from_numpy(x).cuda()
    nb = BYTES[fmt]
    dout = torch.zeros(S * F * Cn * nb, dtype=torch.uint8, device="cuda")
    stats = torch.zeros((S, 2), dtype=torch.float64, device="cuda")
    assert L.dspamd_pcm_write(FMTS[fmt][0], dx.data_ptr(), F, dout.data_ptr(), S, F, Cn, 0, 0, stats.data_ptr(), None) == 0
    torch.cuda.synchronize()
    got = dout.cpu().numpy().reshape(S, -1)
    st = stats.cpu().numpy()
    for s in range(S):
        want, _, ost = orc_write(x[s].reshape(-1), fmt)
        assert np.array_equal(got[s], want)
        assert st[s, 1] == ost[1] and st[s].view(np.uint64)[0] == int(ost[0])
    # and back
    back = torch.zeros(S * F * Cn, dtype=torch.float64, device="cuda")
    assert L.dspamd_pcm_read(FMTS[fmt][0], dout.data_ptr(), back.data_ptr(), S * F * Cn, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(back.cpu().numpy(), orc_read(got.reshape(-1), fmt))


@pytest.mark.gpu
@pytest.mark.parametrize("prec,fmt", [(16, "s16"), (24, "s24_3"), (8, "u8")])
def test_device_dither_is_the_reference_sequence_across_calls(gpu, prec, fmt):
    # the dither generators advance once per sample in interleaved order from the start of the stream: two calls with
    # frames_before set continue the sequence exactly; every stream has its own sequence (its own reference process)
    L, torch = gpu
    S, F, Cn = 2, 7001, 2
    x = np.stack([test_signal(F * Cn, 20 + s, 1.05).reshape(F, Cn) for s in range(S)])
    dx = torch.from_numpy(x).cuda()
    nb = BYTES[fmt]
    cut = 3000
    parts = []
    for f0, f1 in ((0, cut), (cut, F)):
        n = f1 - f0
        dout = torch.zeros(S * n * Cn * nb, dtype=torch.uint8, device="cuda")
        seg = dx[:, f0:f1, :].contiguous()
        assert L.dspamd_pcm_write(FMTS[fmt][0], seg.data_ptr(), n, dout.data_ptr(), S, n, Cn, prec, f0, None, None) == 0
        torch.cuda.synchronize()
        parts.append(dout.cpu().numpy().reshape(S, -1))
    got = np.concatenate(parts, axis=1)
    for s in range(S):
        want, _, _ = orc_write(x[s].reshape(-1), fmt, prec)
        assert np.array_equal(got[s], want)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(DSP_REF), reason="oracle/_ref/dsp_ref not built")
def test_device_sink_equals_the_reference_cli(gpu, tmp_path):
    L, torch = gpu
    F, Cn = 9001, 2
    x = test_signal(F * Cn, 5)
    want = cli_bytes(tmp_path, x, Cn, "s16", True)
    dx = torch.from_numpy(x.reshape(1, F, Cn)).cuda()
    dout = torch.zeros(F * Cn * 2, dtype=torch.uint8, device="cuda")
    assert L.dspamd_pcm_write(2, dx.data_ptr(), F, dout.data_ptr(), 1, F, Cn, 16, 0, None, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(dout.cpu().numpy(), want)
