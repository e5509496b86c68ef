#!/usr/bin/env python3
"""Stress of the resident small-block wave (kernels_resident.hip): chains built, driven and destroyed in quick succession through the reference's chain runtime
(oracle/_ref/libdspref_gpu.so), with pauses around the wave's lifetime, while batch kernels and device-to-host copies run in the same process.
usage: stress_resident.py [seconds=40]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dsp_amd
dsp_amd.load_library()
from oracle_api import RefChain
BIQ = "lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 eq 800 1.0 -1"
CH = ["gain -3 " + BIQ, "remix 0 1 0 1 :0,1 lowpass 2k 0.707 :2,3 highpass 2k 0.707 : gain -1", "gain -6 mult 1.5 add 0.25"]
rng = np.random.default_rng(1)
f = "/tmp/stress_h.raw"
h = rng.standard_normal(3000) * np.exp(-np.arange(3000) / 500.0); np.asarray(h / np.sqrt(np.sum(h * h)) / 4, dtype="<f8").tofile(f)
b = dsp_amd.BatchChain(f"lowpass 1k 0.707 fir_p -t pcm -e double -c 1 {f}", 48000, 2, 64, 65536)
xb = torch.rand((64, 65536, 2), dtype=torch.float64, device="cuda") - 0.5
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 40.0)
n = 0
live = []
while time.time() < t_end:
    chain = CH[n % 3]
    r = RefChain(chain, 48000, 2, variant="_gpu")
    x = rng.uniform(-0.5, 0.5, size=(64 * 40, 2))
    for k in range(40):
        y = r.run(x[64 * k:64 * k + 64])
        if k % 7 == int(rng.integers(7)): time.sleep(float(rng.choice([0.0, 0.001, 0.0029, 0.0031, 0.006, 0.021])))
        if k % 13 == 0:
            yb = b.run(xb); _ = yb[3].cpu()                      # batch kernels and a device-to-host copy beside a live wave
    assert np.isfinite(y).all()
    live.append(r)
    if len(live) > int(rng.integers(1, 4)):                      # a few chains alive at once, closed in random order
        live.pop(int(rng.integers(len(live)))).close()
    n += 1
for r in live: r.close()
torch.cuda.synchronize()
print("chains", n, "ok")
