#!/usr/bin/env python3
"""run() latency of 64-frame blocks arriving in REAL TIME (one block per 1.333 ms, the caller busy-waits in between), not back to back: what a LADSPA host
sees.  Through the reference's chain runtime over libdsp_amd.so; DSP_AMD_PLUGIN_MAILBOX=host|device, DSP_AMD_PLUGIN_RESIDENT=0 for the launch path.
usage: r06_paced_blocks.py [period_us=1333] [blocks=1500]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dsp_amd
dsp_amd.load_library()
from oracle_api import RefChain

BIQ = "lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707"
period = float(sys.argv[1]) * 1e-6 if len(sys.argv) > 1 else 1333e-6
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
variant = os.environ.get("REF_VARIANT", "_gpu")
r = RefChain("gain -3 " + BIQ, 48000, 2, variant=variant)
x = np.random.default_rng(1).uniform(-0.5, 0.5, size=(64 * blocks, 2))
ts = []
nxt = time.perf_counter()
for k in range(blocks):
    while time.perf_counter() < nxt:
        pass
    t0 = time.perf_counter()
    r.run(x[64 * k:64 * k + 64])
    ts.append(time.perf_counter() - t0)
    nxt += period
r.close()
ts = np.array(ts[100:]) * 1e6
print(f"period {period * 1e6:.0f} us, {len(ts)} blocks: median {np.median(ts):.2f} us, mean {ts.mean():.2f}, p99 {np.percentile(ts, 99):.2f}, max {ts.max():.1f}  "
      f"[mailbox {os.environ.get('DSP_AMD_PLUGIN_MAILBOX', 'device')}, resident {os.environ.get('DSP_AMD_PLUGIN_RESIDENT', '1')}, variant '{variant}']")
