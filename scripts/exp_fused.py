#!/usr/bin/env python3
"""The cascade fused into the convolver's first pass (kernels_fused.hip) against the separate kernels and the real reference,
at a small shape (N = 2^18) with error maps, then timed at the headline shape.  usage: python scripts/exp_fused.py [small|time|all]"""
import os, sys, json, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dsp_amd
from oracle_api import RefChain, rms
BIQ = ("lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 eq 800 1.0 -1 "
       "eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707")


def make_filter(n, seed=7, decay=8000.0):
    rng = np.random.default_rng(seed)
    h = rng.standard_normal(n) * np.exp(-np.arange(n) / decay)
    return h / np.sqrt(np.sum(h * h)) / 4.0


def build(chain, C, S, B, fuse):
    os.environ["DSP_AMD_FUSE"] = "1" if fuse else "0"
    b = dsp_amd.BatchChain(chain, 48000, C, S, B)
    os.environ.pop("DSP_AMD_FUSE")
    return b


def small(taps=16384, S=8, C=8, B=245760, biq=BIQ, tag="10 sections"):
    p = f"/tmp/fz_{taps}.raw"
    np.asarray(make_filter(taps), dtype="<f8").tofile(p)
    chain = f"{biq} fir_p -t pcm -e double -c 1 {p}"
    bf, bs = build(chain, C, S, B, True), build(chain, C, S, B, False)
    print(tag, "\n  fused   :", bf.plan(), "\n  separate:", bs.plan(), flush=True)
    assert "cascade-fused" in bf.plan() and "cascade-fused" not in bs.plan()
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    sizes = [B, B, 5000, B]          # the third call leaves the grid: separate kernels on the fused instance too, then fused again
    xs = [torch.rand((S, n, C), dtype=torch.float64, device="cuda", generator=g) - 0.5 for n in sizes]
    yf = [bf.run(x).clone() for x in xs]
    ys = [bs.run(x).clone() for x in xs]
    torch.cuda.synchronize()
    ok = True
    for k, (a, b_) in enumerate(zip(yf, ys)):
        d = (a - b_)
        e = float(d.pow(2).mean().sqrt()); ref = float(b_.pow(2).mean().sqrt())
        print(f"  call {k} ({sizes[k]} frames): rms(fused - separate) = {e:.3e}  (signal {ref:.3e})  max = {float(d.abs().max()):.3e}  finite = {bool(torch.isfinite(a).all())}", flush=True)
        if not (e < 1e-12 * max(ref, 1e-3) * 100):
            ok = False
            # where: per stream / channel, then a (row, column) map of stream 0 channel 0 in window coordinates of the call
            per = d.pow(2).mean(dim=1).sqrt().cpu().numpy()
            print("   per (stream, channel) rms:\n", np.array2string(per, precision=2, max_line_width=200))
            if sizes[k] == B:
                N2 = B // 240
                m = d[0, :, 0].reshape(240, N2).abs().cpu().numpy()
                rows = m.max(axis=1); cols = m.max(axis=0)
                print("   stream 0 ch 0: bad rows", np.nonzero(rows > 1e-9)[0][:40], "bad cols", np.nonzero(cols > 1e-9)[0][:40])
    if RefChain.available():
        for s in (0, S - 1):
            x = torch.cat([t[s] for t in xs], dim=0).cpu().numpy()
            ref = RefChain(chain, 48000, C).run(x)
            for name, ys_ in (("fused", yf), ("separate", ys)):
                got = torch.cat([t[s] for t in ys_], dim=0).cpu().numpy()
                print(f"  stream {s}: rms({name} - reference) = {rms(ref - got):.3e}", flush=True)
                if name == "fused" and not rms(ref - got) < 1e-12: ok = False
    # reset + rerun gives the same bits
    bf.reset()
    y0 = bf.run(xs[0]).clone()
    print("  reset / rerun bit-identical:", bool(torch.equal(y0, yf[0])), flush=True)
    print("  SMALL", "OK" if ok else "FAILED", flush=True)
    return ok


def timed(S=256, C=8, B=983040, taps=65536, steps=6, modes=(True, False, True)):
    L = dsp_amd.load_library()
    p = f"/tmp/fz_{taps}.raw"
    np.asarray(make_filter(taps), dtype="<f8").tofile(p)
    chain = f"{BIQ} fir_p -t pcm -e double -c 1 {p}"
    res = {}
    x = torch.zeros((S, B + 68, C), dtype=torch.float64, device="cuda")
    L.dspamd_sgen_sine(x.data_ptr(), S, B + 68, C, 48000, 100.0, 90.0, 0, torch.cuda.current_stream().cuda_stream)
    o = torch.empty((S, B + 68, C), dtype=torch.float64, device="cuda")
    outs = {}
    for fuse in modes:
        if isinstance(fuse, str):
            os.environ["DSP_AMD_FUSE_DBG"] = fuse; tagname = "dbg" + fuse; fuse = True
        else:
            os.environ.pop("DSP_AMD_FUSE_DBG", None); tagname = "fused" if fuse else "separate"
        b = build(chain, C, S, B, fuse)
        for _ in range(2): b.run(x[:, :B, :], o)
        torch.cuda.synchronize()
        L.dspamd_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(steps): b.run(x[:, :B, :], o)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        prof = {}
        for line in L.dspamd_profile_collect().decode().splitlines():
            name, ms, cnt = line.split()
            prof[name] = round(float(ms) / max(int(cnt), 1), 3)
        L.dspamd_profile_enable(0)
        outs[fuse] = o[:, :B, :].clone()
        res[tagname] = {"ms_per_step": round(dt * 1e3, 3), "Gsamples_s": round(S * C * B / dt / 1e9, 2), "kernels_ms": prof}
        print(tagname, json.dumps(res[tagname]), flush=True)
        del b
    if False not in outs: return res
    d = outs[True] - outs[False]
    print("headline shape, third step: rms(fused - separate) =", float(d.pow(2).mean().sqrt()), "signal", float(outs[False].pow(2).mean().sqrt()), flush=True)
    return res


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.cuda.set_device(0)
    if what in ("small", "all"):
        ok = small()
        ok2 = small(taps=32768, S=8, C=4, B=491520, biq="lowpass 1k 0.707 gain -3 eq 400 2.0 1.5 gain 2", tag="2 sections + gains, 4 ch, N = 2^19")
    if what == "dbg":
        timed(steps=3, modes=[True] + sys.argv[2:])
    if what in ("time", "all"):
        r = timed()
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(r, open(os.path.join(ROOT, "gpurun_out", "exp_fused.json"), "w"), indent=1)
