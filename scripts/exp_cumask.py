#!/usr/bin/env python3
"""Experiment: the batch in two halves on two HIP streams with DISJOINT CU masks (hipExtStreamCreateWithCUMask), free running:
one half's cascade (fp64-bound) next to the other half's transforms (HBM-bound), neither able to take the other's CUs.
Prints ms per whole step (both halves) against the single-batch step."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dsp_amd
from bench import BIQUADS, make_filter

S, Cn, B, taps, fs = 256, 8, 983040, 65536, 48000
d = f"/tmp/cum_{os.getpid()}"; os.makedirs(d, exist_ok=True)
np.asarray(make_filter(taps), dtype="<f8").tofile(d + "/filt.raw")
chain = BIQUADS + " fir_p -t pcm -e double -c 1 filt.raw"
L = dsp_amd.load_library()
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))

def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    st = C.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words) == 0
    return torch.cuda.ExternalStream(st.value)

def mk(S):
    b = dsp_amd.BatchChain(chain, fs, Cn, S, B, directory=d)
    x = torch.zeros((S, B + 68, Cn), dtype=torch.float64, device="cuda")
    x[:, :B, :] = torch.rand((S, B, Cn), dtype=torch.float64, device="cuda") - 0.5
    o = torch.empty((S, B + 68, Cn), dtype=torch.float64, device="cuda")
    return b, x[:, :B, :], o

def free_running(halves, streams, steps=8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for (b, x, o), st in zip(halves, streams):
        with torch.cuda.stream(st):
            for _ in range(steps): b.run(x, o)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3

whole = mk(S)
for _ in range(2): whole[0].run(whole[1], whole[2])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): whole[0].run(whole[1], whole[2])
torch.cuda.synchronize(); print("single batch of %d streams: %.3f ms per step" % (S, (time.perf_counter() - t0) / 8 * 1e3))
del whole; torch.cuda.empty_cache()
halves = [mk(S // 2) for _ in range(2)]
ALL = (1 << 256) - 1
EVEN = int("55" * 32, 16); ODD = ALL ^ EVEN
LO = (1 << 128) - 1; HI = ALL ^ LO
XA = int("0f" * 32, 16); XB = ALL ^ XA            # nibbles: 4 CUs on, 4 off
for name, ma, mb in (("no masks", ALL, ALL), ("even / odd CUs", EVEN, ODD), ("low / high half", LO, HI), ("alternating groups of 4", XA, XB)):
    sts = [masked_stream(ma), masked_stream(mb)]
    free_running(halves, sts, 2)
    print("2 x %d streams, %s: %.3f ms per step" % (S // 2, name, free_running(halves, sts, 8)), flush=True)
