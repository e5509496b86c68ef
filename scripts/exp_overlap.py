#!/usr/bin/env python3
"""Experiment: does a VALU-bound kernel (the cascade of one half of the streams) overlap with the HBM-bound transform kernels of the
other half when the two halves run on two HIP streams?  Two BatchChain objects of S / 2 streams each, staggered with events so
that the two cascades never run at the same time.  Prints ms per whole step (both halves) against the single-batch step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dsp_amd
from bench import BIQUADS, make_filter

S, C, B, taps, fs = 256, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 983040, 65536, 48000
d = f"/tmp/ovl_{os.getpid()}"; os.makedirs(d, exist_ok=True)
np.asarray(make_filter(taps), dtype="<f8").tofile(d + "/filt.raw")
chain = BIQUADS + " fir_p -t pcm -e double -c 1 filt.raw"
L = dsp_amd.load_library()

def mk(S):
    b = dsp_amd.BatchChain(chain, fs, C, S, B, directory=d)
    x = torch.zeros((S, B + 68, C), dtype=torch.float64, device="cuda")
    x[:, :B, :] = torch.rand((S, B, C), dtype=torch.float64, device="cuda") - 0.5
    o = torch.empty((S, B + 68, C), dtype=torch.float64, device="cuda")
    return b, x[:, :B, :], o

def timeit(fn, steps=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3

whole = mk(S)
print("single batch of %d streams: %.3f ms per step" % (S, timeit(lambda: whole[0].run(whole[1], whole[2]))))
del whole; torch.cuda.empty_cache()
for parts in (2, 4):
    halves = [mk(S // parts) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def step_plain():
        for (b, x, o), st in zip(halves, streams):
            with torch.cuda.stream(st): b.run(x, o)
    print("%d x %d streams on %d HIP streams, no staggering: %.3f ms per step" % (parts, S // parts, parts, timeit(step_plain)))
    # staggered: the second HIP stream starts a third of a step late (a device-side sleep), so that its cascade meets the other
    # half's transforms; nothing re-synchronises the two afterwards
    def run_staggered(steps=8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k, ((b, x, o), st) in enumerate(zip(halves, streams)):
            with torch.cuda.stream(st):
                if k: torch.cuda._sleep(int(2.0e9 * 0.010 * k / parts))
                for _ in range(steps): b.run(x, o)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
    run_staggered(2)
    print("%d x %d streams on %d HIP streams, staggered start: %.3f ms per step (8 steps back to back per stream)" % (parts, S // parts, parts, run_staggered(8)))
    def run_back_to_back(steps=8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for (b, x, o), st in zip(halves, streams):
            with torch.cuda.stream(st):
                for _ in range(steps): b.run(x, o)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
    print("%d x %d streams on %d HIP streams, free running: %.3f ms per step" % (parts, S // parts, parts, run_back_to_back(8)))
    def step_serial():
        for (b, x, o) in halves: b.run(x, o)
    print("%d x %d streams on ONE HIP stream: %.3f ms per step" % (parts, S // parts, timeit(step_serial)))
