#!/usr/bin/env python3
"""Soak of the one-trip convolver (kernels_short.hip): random filters of 17 ... 8193 taps (one shared or one per channel, `fir` or `fir_p`, with and without a
selector, a cascade in front, a consumer behind), random shapes and call sequences (whole multiples of the hop, ragged sizes, single frames, resets), every
output compared with the four-step transforms (DSP_AMD_CONV_SHORT=0) on the same inputs.  usage: soak_short.py [seeds=40]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dsp_amd


def build(chain, C, S, B, short):
    if short is not True: os.environ["DSP_AMD_CONV_SHORT"] = str(int(short))     # False: the four-step transforms; 13 / 14: that window of the one-trip form
    try: return dsp_amd.BatchChain(chain, 48000, C, S, B)
    finally: os.environ.pop("DSP_AMD_CONV_SHORT", None)


def main(n_seeds):
    worst = 0.0
    for seed in range(n_seeds):
        rng = np.random.default_rng(5000 + seed)
        C = int(rng.choice([1, 2, 3, 4, 8])); S = int(rng.choice([1, 2, 7, 33, 130]))
        taps = int(rng.choice([17, 33, 100, 1000, 2049, 4000, 4095, 4096, 4097, 4098, 6000, 8192, 8193, int(rng.integers(17, 4098)), int(rng.integers(4098, 8194))]))
        per_ch = C > 1 and rng.random() < 0.25
        h = rng.standard_normal((taps, C if per_ch else 1)) * np.exp(-np.arange(taps) / 500.0)[:, None]
        h = h / np.sqrt(np.sum(h * h, axis=0)) / 4
        f = f"/tmp/soak_short_{seed}.raw"; np.asarray(h, dtype="<f8").tofile(f)
        eff = "fir" if rng.random() < 0.3 else "fir_p"
        sel = ""
        if C >= 2 and not per_ch and rng.random() < 0.3: sel = ":" + ",".join(str(c) for c in sorted(rng.choice(C, size=int(rng.integers(1, C)), replace=False)))
        pre = "lowpass 2k 0.707 eq 300 1.5 4 " if rng.random() < 0.3 else ""
        post = " gain -1.5 highshelf 6k 0.7 2" if rng.random() < 0.3 else ""
        chain = f"{pre}{sel} {eff} -t pcm -e double -c {C if per_ch else 1} {f}{' :' if sel else ''}{post}".strip()
        B = int(rng.choice([1024, 4096, 20000, 65536, 49152]))
        window = [True, 13, 14][int(rng.integers(0, 3))]
        bo, bs = build(chain, C, S, B, window), build(chain, C, S, B, False)
        one_trip = "one-trip" in bo.plan()
        g = torch.Generator(device="cuda"); g.manual_seed(seed)
        err = 0.0
        for step in range(6):
            r = rng.random()
            n = B if r < 0.4 else int(rng.choice([1, 8 * int(rng.integers(1, B // 8 + 1)), int(rng.integers(1, B + 1))]))
            if rng.random() < 0.1: bo.reset(); bs.reset()
            x = torch.rand((S, n, C), dtype=torch.float64, device="cuda", generator=g) - 0.5
            a, b = bo.run(x).clone(), bs.run(x).clone()
            assert a.shape == b.shape and bool(torch.isfinite(a).all()), (chain, a.shape, b.shape)
            if a.numel(): err = max(err, float((a - b).abs().max()))
        while True:
            a, b = bo.drain(B), bs.drain(B)
            assert (a is None) == (b is None)
            if a is None: break
            assert a.shape == b.shape
            if a.numel(): err = max(err, float((a - b).abs().max()))
        print(f"seed {seed}: S {S} C {C} taps {taps}{' per-channel' if per_ch else ''} {eff} sel '{sel}' pre {bool(pre)} post {bool(post)} block {B} window {window} one-trip {one_trip} {[w for w in bo.plan().split() if w.startswith('N=')][:1]}  max |one-trip - four-step| = {err:.2e}", flush=True)
        worst = max(worst, err)
        os.remove(f)
        del bo, bs
    print("worst", worst)
    assert worst < 1e-12


if __name__ == "__main__":
    torch.cuda.set_device(0)
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
