#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/libdspref.so,
built by oracle/Makefile from /root/reference).  Run in the authoring container:

    make -C oracle && python tests/golden/make_golden.py

The reference ships no golden vectors of its own; these
fixtures are outputs of the reference itself on seeded inputs, frozen so that
the oracle and the HIP path can be checked where /root/reference is absent.
Inputs are regenerated from the seeds recorded in each case (PCG64 uniform).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_api import RefChain  # noqa: E402

CONFIG2 = ("lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 "
           "eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707")


def noise(frames, ch, seed, amp=0.5):
    return np.random.Generator(np.random.PCG64(seed)).uniform(-amp, amp, size=(frames, ch))


def make_filter(n, seed=7, decay=None):
    rng = np.random.default_rng(seed)
    h = rng.standard_normal(n) * np.exp(-np.arange(n) / (decay or max(n / 8.0, 1.0)))
    return h / np.sqrt(np.sum(h * h)) / 4.0


def main():
    cases = []
    arrays = {}

    def add(name, chain, fs, ch, frames, seed, block, filt=None, amp=0.5, oracle=True):
        x = noise(frames, ch, seed, amp)
        c = chain
        if filt is not None:
            path = os.path.join("/tmp", f"golden_{name}.raw")
            np.asarray(filt, dtype="<f8").tofile(path)
            c = chain.replace("{F}", path)
            arrays[f"{name}__filter"] = np.asarray(filt)
        rc = RefChain(c, fs, ch)
        y = rc.process(x, block=block)
        arrays[f"{name}__out"] = y
        cases.append(dict(name=name, chain=chain, fs=fs, channels=ch, frames=frames, seed=seed, amp=amp,
                          block=block, ofs=rc.ofs, ochannels=rc.ochannels, effects=rc.effect_names(), oracle=oracle))

    # impulse responses -> coefficient known answers for every biquad type
    biquads = ["lowpass_1 1k", "highpass_1 300", "allpass_1 2k", "lowshelf_1 200 4", "highshelf_1 5k -3",
               "lowpass_1p 800", "lowpass 1k 0.707", "highpass 20 0.707", "bandpass_skirt 1k 2",
               "bandpass_peak 1k 1o", "notch 60 10", "allpass 500 200h", "eq 100 1.0 3", "eq 3200 1k -2.5",
               "lowshelf 100 0.7 6", "lowshelf 100 0.8s 6", "highshelf 8k 0.7 -3", "highshelf 8k 6d -3",
               "lowpass_transform 80 0.9 40 0.5", "linkwitz_transform 80 0.9 40 0.5", "deemph",
               "biquad 0.2 0.3 0.1 1.1 -0.4 0.2", "lowpass 2k bw4.1"]
    imp = np.zeros((48, 1)); imp[0] = 1.0
    ir = {}
    for b in biquads:
        ir[b] = RefChain(b, 48000, 1).process(imp, block=48)[:, 0]
    arrays["biquad_ir"] = np.stack([ir[b] for b in biquads])

    add("config1", "gain -6 lowpass 1k 0.707", 48000, 2, 600, 1, 256)
    add("config2", CONFIG2, 48000, 8, 1024, 1234, 300)
    add("gain_sel", "gain -6 :1,3 mult 0.3 : add 0.001", 48000, 4, 300, 2, 128)
    add("remix", "remix 0,1 2 . 1,2,3", 48000, 4, 300, 3, 128)
    add("remix_up", "remix 0 1 0,1", 48000, 2, 300, 4, 128)
    add("delay", ":1 delay 37S", 48000, 2, 300, 5, 100)
    add("fir_direct", "fir coefs:0.5,-0.25,0.125,0.0625/0.1,0.2,0.3,0.4", 48000, 2, 300, 6, 64)
    add("fir_p_1000", "fir_p -t pcm -e double -c 1 {F}", 48000, 2, 1500, 7, 512, filt=make_filter(1000))
    add("fir_p_5000", "fir_p -t pcm -e double -c 1 {F}", 48000, 1, 3000, 8, 1000, filt=make_filter(5000, seed=8))
    add("fir_100", "fir -t pcm -e double -c 1 {F}", 48000, 2, 1000, 9, 256, filt=make_filter(100, seed=9))
    add("resample_2x", "resample 96k", 48000, 2, 900, 10, 300, amp=0.4)
    add("resample_half", "resample 48k", 96000, 2, 1800, 11, 500, amp=0.4)
    add("resample_441_48", "resample 48k", 44100, 1, 900, 12, 300, amp=0.4)
    add("hilbert_p255", "hilbert -p 255", 48000, 1, 800, 13, 256)
    add("chain4", "gain -3 lowpass 1k 0.707 eq 400 2.0 1.5 fir_p -t pcm -e double -c 1 {F} resample 96k", 48000, 2, 1200, 14, 400,
        filt=make_filter(700, seed=14), amp=0.4)

    # checked against the real reference only (oracle=False: the C restatement does not cover them)
    add("riir", "lowpass 2k 0.707 lowpass -r 2k 0.707 :0 highpass -r60 30 0.707", 48000, 2, 3000, 15, 1000, oracle=False)
    add("delay_frac", ":0 delay -f 0.37S :1 delay -f5 7.3S : eq 500 1.0 2", 48000, 2, 1200, 16, 500)
    add("midside", "st2ms :1 mult 0.5 : ms2st", 48000, 2, 600, 17, 256)
    add("crossfeed", "crossfeed 700 4.5", 48000, 2, 1500, 18, 512)
    add("fir_p_ragged", "fir_p coefs:0.5,0.25,-0.125,0.0625,0.03,0.01,0.5,0.25,-0.125,0.0625,0.03,0.01,0.5,0.25,-0.125,0.0625,0.03,0.01,0.2,0.1,"
        "0.5,0.25,-0.125,0.0625,0.03,0.01,0.5,0.25,-0.125,0.0625,0.03,0.01,0.5,0.25,-0.125,0.0625,0.03,0.01,0.2,0.1/0.9,-0.3,0.1", 48000, 2, 900, 19, 300, oracle=False)

    np.savez_compressed(os.path.join(HERE, "golden.npz"), **arrays)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(dict(biquads=biquads, cases=cases), f, indent=1)
    print("wrote", len(cases), "cases;", os.path.getsize(os.path.join(HERE, "golden.npz")), "bytes")


if __name__ == "__main__":
    main()
