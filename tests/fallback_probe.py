"""Not a test module: `python tests/fallback_probe.py out.npz` runs a fixed battery of small cases through the library (shapes
picked so that every kernel family is reached) and saves the outputs.  tests/test_gpu_fallbacks.py runs it once per DSP_AMD_*
switch in a process of its own (the switches are read once per process) and holds every run to the reference's outputs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BIQ = ("lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 "
       "eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707")


def noise(n, ch, seed, amp=0.4):
    return np.random.Generator(np.random.PCG64(seed)).uniform(-amp, amp, size=(n, ch))


def make_filter(taps, seed, decay):
    rng = np.random.Generator(np.random.PCG64(seed))
    h = rng.standard_normal(taps) * np.exp(-np.arange(taps) / decay)
    return h / np.sqrt(np.sum(h * h)) / 4.0


# name -> dict(chain, S, C, frames, calls, pick (streams compared), taps=(n, seed, decay) or None, log2n, kind)
CASES = {
    "rows4":    dict(chain="gain -2 " + BIQ, S=128, C=8, frames=5000, calls=(2048, 2952), pick=(0, 77, 127)),
    "rows2":    dict(chain=BIQ, S=64, C=8, frames=5000, calls=(3072, 1928), pick=(0, 63)),
    "rows1":    dict(chain=BIQ, S=16, C=8, frames=6000, calls=(4096, 1904), pick=(0, 15)),
    "mixed":    dict(chain="lowpass 1k 0.707 :0,3 eq 400 2.0 1.5 : add 0.001 highshelf 8k 0.7 -3", S=24, C=4, frames=4000, calls=(2048, 1952), pick=(0, 23)),
    "chunk":    dict(chain=BIQ, S=1, C=8, frames=65536, calls=(32768, 32768), pick=(0,)),
    "conv4096": dict(chain="fir_p -t pcm -e double -c 1 {F}", S=16, C=2, frames=9000, calls=(2500, 2500, 2500, 1500), pick=(0, 15), taps=(3001, 5, 300.0), log2n=20),
    "conv2048": dict(chain="fir_p -t pcm -e double -c 1 {F}", S=16, C=2, frames=9000, calls=(2500, 2500, 2500, 1500), pick=(0, 15), taps=(3001, 5, 300.0), log2n=19),
    "conv1024": dict(chain="fir_p -t pcm -e double -c 1 {F}", S=16, C=2, frames=9000, calls=(2500, 2500, 2500, 1500), pick=(3, 15), taps=(3001, 5, 300.0), log2n=18),
    "conv512":  dict(chain="fir_p -t pcm -e double -c 1 {F}", S=8, C=8, frames=9000, calls=(4500, 4500), pick=(0, 7), taps=(1500, 6, 200.0), log2n=13),
    "conv_k3p": dict(chain="fir_p -t pcm -e double -c 1 {F}", S=8, C=8, frames=9000, calls=(4500, 4500), pick=(0, 7), taps=(3001, 13, 300.0), log2n=18),   # four pairs per stream, >= 1024 tiles: the persistent K3
    "zita":     dict(chain="zita_convolver -t pcm -e double -c 1 {F}", S=12, C=2, frames=7000, calls=(3500, 3500), pick=(0, 11), taps=(3001, 14, 300.0), log2n=19),   # float32-spectrum instance (fp64 transforms with DSP_AMD_ZITA_F64=1)
    "headline": dict(chain=BIQ + " fir_p -t pcm -e double -c 1 {F}", S=8, C=8, frames=12000, calls=(4096, 4096, 3808), pick=(0, 7), taps=(5000, 7, 600.0)),
    "fir_lat":  dict(chain="fir -t pcm -e double -c 1 {F} :1 delay 7S", S=3, C=3, frames=6000, calls=(3000, 3000), pick=(0, 2), taps=(700, 8, 90.0)),
    "two_conv": dict(chain="fir_p -t pcm -e double -c 1 {F} hilbert -p 255", S=4, C=2, frames=7000, calls=(3500, 3500), pick=(0, 3), taps=(900, 9, 120.0)),
    "rs96":     dict(chain=BIQ + " fir_p -t pcm -e double -c 1 {F} resample 96k", S=3, C=8, frames=9000, calls=(4096, 4904), pick=(0, 2), taps=(2000, 10, 300.0)),
    "rs96_duo": dict(chain="fir_p -t pcm -e double -c 1 {F} resample 96k", S=8, C=2, frames=9000, calls=(4096, 4904), pick=(0, 7), taps=(2000, 10, 300.0), log2n=19),   # 2048-point rows, 8 pairs: the two-branch two-workgroup K2 (round 5; DSP_AMD_ROW_DUO2=0: the persistent three-pass kernel)
    "rs441":    dict(chain="resample 44.1k", S=2, C=2, frames=9000, calls=(4000, 5000), pick=(0, 1)),
    "rs32":     dict(chain="resample 32k", S=2, C=3, frames=6000, calls=(6000,), pick=(0, 1)),
    "small":    dict(chain="fir_p -t pcm -e double -c 1 {F}", S=4, C=2, frames=24576, calls=(2048,) * 12, pick=(0, 3), taps=(40000, 11, 6000.0)),
    "mid":      dict(chain="fir_p -t pcm -e double -c 1 {F}", S=4, C=2, frames=57344, calls=(8192,) * 7, pick=(0, 3), taps=(40000, 12, 6000.0)),   # 5 slots of 8192 taps in the row kernel's delay line (mid-size calls)
    "remix":    dict(chain="remix 0,1 2 . 1,2,3 :0 delay 37S", S=5, C=4, frames=3000, calls=(1000, 2000), pick=(0, 4)),
    # calls of exactly one hop of a 256 x 1024 transform whose history is 32 whole rows: the cascade fused into the convolver's first
    # pass (kernels_fused.hip; the separate kernels with DSP_AMD_FUSE=0; 8 channels: the chunks' end states on the
    # matrix cores, or by the recurrence where the shape is not served), then a call off the grid, which the separate kernels take
    "fused":    dict(chain="gain -1.5 " + BIQ + " fir_p -t pcm -e double -c 1 {F}", S=2, C=8, frames=2 * 229376 + 3000, calls=(229376, 229376, 3000), pick=(0, 1), taps=(32768, 15, 4000.0)),
}
HOST_CASES = {   # through dspamd_chain_run (host buffers: mapped staging / copy commands)
    "host_eq":   dict(chain="gain -3 " + BIQ, C=2, frames=9000, block=512),
    "host_big":  dict(chain="gain -3 " + BIQ, C=8, frames=24000, block=8192),     # 512 KB per block: page-locked staging / copy commands
    "host_conv": dict(chain="fir_p -t pcm -e double -c 1 {F} resample 44.1k", C=2, frames=9000, block=2048, taps=(800, 12, 100.0)),
    "host_huge": dict(chain="gain -3 " + BIQ, C=2, frames=300000, block=131072),   # blocks of two pipeline calls: page-locked double buffers filled by the caller and its helper threads (DSP_AMD_COPY_CREW=0: the caller alone)
}


def inputs(name, c):
    seed = sum(ord(ch) for ch in name)
    if name in HOST_CASES:
        return noise(c["frames"], c["C"], seed)
    return np.stack([noise(c["frames"], c["C"], seed * 131 + s) for s in range(c["S"])])


def filter_of(c):
    return make_filter(*c["taps"]) if c.get("taps") else None


def main(out_path, only=None):
    import torch
    import dsp_amd
    res = {}
    tmp = f"/tmp/dsp_amd_probe_{os.getpid()}"
    os.makedirs(tmp, exist_ok=True)
    for name, c in CASES.items():
        if only and name not in only:
            continue
        chain = c["chain"]
        h = filter_of(c)
        if h is not None:
            f = os.path.join(tmp, f"{name}.raw")
            np.asarray(h, dtype="<f8").tofile(f)
            chain = chain.replace("{F}", f)
        if c.get("log2n"):
            os.environ["DSP_AMD_CONV_LOG2N"] = str(c["log2n"])
        else:
            os.environ.pop("DSP_AMD_CONV_LOG2N", None)
        x = inputs(name, c)
        b = dsp_amd.BatchChain(chain, 48000, c["C"], c["S"], max(c["calls"]))
        xt = torch.from_numpy(x).cuda()
        outs, pos = [], 0
        for n in c["calls"]:
            outs.append(b.run(xt[:, pos:pos + n, :].contiguous()).clone())
            pos += n
        while True:
            o = b.drain(max(c["calls"]))
            if o is None:
                break
            outs.append(o.clone())
        y = torch.cat([o for o in outs if o.shape[1]], dim=1).cpu().numpy()
        for s in c["pick"]:
            res[f"{name}/{s}"] = y[s]
        res[f"{name}/plan"] = np.array(b.plan())
        del b
    os.environ.pop("DSP_AMD_CONV_LOG2N", None)
    for name, c in HOST_CASES.items():
        if only and name not in only:
            continue
        chain = c["chain"]
        h = filter_of(c)
        if h is not None:
            f = os.path.join(tmp, f"{name}.raw")
            np.asarray(h, dtype="<f8").tofile(f)
            chain = chain.replace("{F}", f)
        res[f"{name}/0"] = dsp_amd.EffectsChain(chain, 48000, c["C"]).process(inputs(name, c), block=c["block"])
    # one wire-format case: s16 -> s16 with dither through the rows cascade, bytes + statistics
    if not only or "wire" in only:
        S, C, F = 64, 8, 4096
        x16 = torch.from_numpy(np.round(noise(S * F, C, 4711, 0.9).reshape(S, F, C) * 32767).astype(np.int16)).cuda()
        b = dsp_amd.BatchChain("gain 4 " + BIQ, 48000, C, S, F)
        st = torch.zeros((S, 2), dtype=torch.float64, device="cuda")
        parts = [b.run_wire(x16[:, :2048, :].contiguous(), "s16", "s16", 16, st).clone(), b.run_wire(x16[:, 2048:, :].contiguous(), "s16", "s16", 16, st).clone()]
        y = torch.cat(parts, dim=1).cpu().numpy()
        res["wire/bytes"] = y[[0, 31, 63]]
        res["wire/stats"] = st.cpu().numpy()
        res["wire/fused"] = np.array(b.wire_fused())
    np.savez(out_path, **res)


if __name__ == "__main__":
    main(sys.argv[1], set(sys.argv[2:]) or None)
