#!/usr/bin/env python3
"""run() time at LADSPA block sizes for segment shapes the resident wave takes since round 6 (remixes / mixes, direct FIRs, up to two cascades), through the
reference's chain runtime over libdsp_amd.so, against a launch per block (DSP_AMD_PLUGIN_RESIDENT=0 in a process of its own) and the all-CPU reference.
usage: r06_wave_shapes_rate.py [frames=64]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
BIQ = "lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707"
FIR16 = "fir_p coefs:" + ",".join(f"{0.9 * (-0.6) ** k:.6f}" for k in range(16))
SHAPES = {
    "equaliser (10 sections)": "gain -3 " + BIQ,
    "crossover 2 -> 4 (remix + 2 sections per band)": "remix 0 1 0 1 :0,1 lowpass 2k 0.707 lowpass 2k 0.707 :2,3 highpass 2k 0.707 highpass 2k 0.707 : gain -1",
    "crossover + 16-tap correction FIR": "remix 0 1 0 1 :0,1 lowpass 2k 0.707 lowpass 2k 0.707 :2,3 highpass 2k 0.707 highpass 2k 0.707 : " + FIR16,
    "16-tap FIR alone": FIR16,
    "FIR + equaliser + gain (two cascades)": "highpass 30 0.707 " + FIR16 + " eq 1k 1.0 3 gain -1",
    "mid/side equaliser (st2ms eq ms2st)": "st2ms eq 1k 1.0 3 eq 4k 1.0 -2 ms2st",
    "crossfeed": "crossfeed 700 4.5",
    # what the wave does NOT take (an FFT convolver in the segment): the launch path whatever the switch says
    "equaliser + 4095-tap fir_p (launches)": "gain -3 lowpass 1k 0.707 eq 400 2.0 1.5 fir_p -t pcm -e double -c 1 /tmp/r06_shapes_h4095.raw",
    "65536-tap fir_p (launches)": "fir_p -t pcm -e double -c 1 /tmp/r06_shapes_h65536.raw",
}
for _t in (4095, 65536):
    _h = np.random.default_rng(_t).standard_normal(_t) * np.exp(-np.arange(_t) / (_t / 6.0))
    np.asarray(_h / np.sqrt(np.sum(_h * _h)) / 4, dtype="<f8").tofile(f"/tmp/r06_shapes_h{_t}.raw")


def measure(variant, frames):
    import dsp_amd
    dsp_amd.load_library()
    from oracle_api import RefChain
    x = np.random.default_rng(1).uniform(-0.5, 0.5, size=(frames * 4000, 2))
    out = {}
    for name, chain in [("warm-up", "gain -1 eq 1k 1.0 1")] + list(SHAPES.items()):
        r = RefChain(chain, 48000, 2, variant=variant)
        for p in range(0, frames * 200, frames):
            r.run(x[p:p + frames])
        t0 = time.perf_counter()
        n = 0
        for p in range(frames * 200, x.shape[0], frames):
            r.run(x[p:p + frames])
            n += 1
        out[name] = (time.perf_counter() - t0) / n * 1e6
        r.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2:
        res = measure(sys.argv[2], int(sys.argv[1]))
        print("RESULT " + repr(res))
        sys.exit(0)
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    legs = {}
    for label, variant, env in (("wave", "_gpu", {}), ("launch per block", "_gpu", {"DSP_AMD_PLUGIN_RESIDENT": "0"}), ("CPU reference", "", {})):
        r = subprocess.run([sys.executable, __file__, str(frames), variant], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        legs[label] = eval(line[0][7:]) if line else {"error": r.stderr[-300:]}
    print(f"# microseconds per run() of {frames} stereo frames, through the reference's chain runtime (its own buffer copies included): scripts/r06_wave_shapes_rate.py")
    print(f"{'segment':52s} {'wave':>8s} {'launch':>8s} {'CPU':>8s}")
    for name in SHAPES:
        print(f"{name:52s} " + " ".join(f"{legs[l].get(name, float('nan')):8.2f}" for l in legs))
