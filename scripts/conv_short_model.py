#!/usr/bin/env python3
"""Model of the one-trip convolver's transform (dsp_amd/csrc/kernels_short.hip, short_fft2): the index arithmetic of its three passes -- radix 32 / 16 / 16 at
8192 points, 32 / 32 / 16 at 16384, a thread holding the 32 points j + (N / 32) m in every pass -- checked against numpy's FFT, and the exchange buffer's slots
(pos + (pos >> 5)) checked for bank conflicts over every store and gather shape (a ds_read / ds_write_b64 is served in two groups of 32 lanes: conflict-free
when the 32 slots differ mod 32).  Runs on the CPU in a few seconds; written before the kernel ran."""
import numpy as np


def slot(pos):
    return pos + (pos >> 5)


def transform(log2n, inv=False):
    N = 1 << log2n; NTH = N // 32; H = NTH; P = N // 16
    RB = 32 if log2n == 14 else 16
    rng = np.random.default_rng(1)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    sgn = +1 if inv else -1
    w = lambda M, e: np.exp(sgn * 2j * np.pi * e / M)
    dft16 = lambda v: np.array([sum(v[n] * w(16, n * r) for n in range(16)) for r in range(16)])

    def dft32(va, vb):
        A, B = dft16(va), dft16(vb) * np.array([w(32, k2) for k2 in range(16)])
        return A + B, A - B
    lds = np.zeros(N + N // 32, complex)
    lds[[slot(p) for p in range(N)]] = x
    gather = lambda j: (np.array([lds[slot(j) + (P + P // 32) * m] for m in range(16)]), np.array([lds[slot(j) + (H + H // 32) + (P + P // 32) * m] for m in range(16)]))
    # pass 1: radix 32, stride 1; stores at 33 j + r
    out = np.zeros_like(lds)
    for j in range(NTH):
        va, vb = dft32(*gather(j))
        for r in range(16):
            out[33 * j + r] = va[r]; out[33 * j + 16 + r] = vb[r]
    lds = out; out = np.zeros_like(lds)
    # middle pass: stride 32, twiddles from the table [r][k]
    for j in range(NTH):
        va, vb = gather(j)
        k = j & 31
        if RB == 16:
            tw = np.array([w(512, r * k) for r in range(16)])
            va, vb = dft16(va * tw), dft16(vb * tw)
            s = (33 * 16) * (j >> 5) + k
            for r in range(16):
                out[s + 33 * r] = va[r]; out[s + (33 * 16) * (H // 32) + 33 * r] = vb[r]
        else:
            va = va * np.array([w(1024, 2 * m * k) for m in range(16)]); vb = vb * np.array([w(1024, (2 * m + 1) * k) for m in range(16)])
            va, vb = dft32(va, vb)
            s = (33 * 32) * (j >> 5) + k
            for r in range(16):
                out[s + 33 * r] = va[r]; out[s + 33 * 16 + 33 * r] = vb[r]
    lds = out
    # last pass: radix 16 per set, stride N / 16, in place; twiddles = powers of W_N^jv
    res = np.zeros(N, complex)
    for j in range(NTH):
        va, vb = gather(j)
        for v, jv in ((va, j), (vb, j + H)):
            u = dft16(v * np.array([w(N, jv) ** r for r in range(16)]))
            for r in range(16):
                res[jv + P * r] = u[r]
    ref = np.fft.ifft(x) * N if inv else np.fft.fft(x)
    return np.abs(res - ref).max() / np.abs(ref).max()


def conflicts(log2n):
    N = 1 << log2n; NTH = N // 32; H = NTH; P = N // 16
    RB = 32 if log2n == 14 else 16
    worst = 1
    for g in range(0, NTH, 32):
        lanes = range(g, g + 32)
        shapes = [[33 * j + r for j in lanes] for r in range(32)]
        shapes += [[slot(j) + (H + H // 32) * m for j in lanes] for m in range(32)]
        if RB == 16:
            shapes += [[(33 * 16) * ((j + H * t) >> 5) + (j & 31) + 33 * r for j in lanes] for t in range(2) for r in range(16)]
        else:
            shapes += [[(33 * 32) * (j >> 5) + (j & 31) + 33 * r for j in lanes] for r in range(32)]
        for sh in shapes:
            assert max(sh) < N + N // 32
            worst = max(worst, 32 // len({a & 31 for a in sh}) if len({a & 31 for a in sh}) else 32)
            assert len({a & 31 for a in sh}) == 32, sh
    # every position has a slot of its own, and the stores of a pass cover exactly the slots the gathers read
    assert len({slot(p) for p in range(N)}) == N
    return worst


if __name__ == "__main__":
    for L in (13, 14):
        print(f"N = {1 << L}: forward {transform(L):.1e}, inverse {transform(L, True):.1e} of the largest bin; worst bank conflict {conflicts(L)}-way")
