"""CPU-only: `plot` (effect.h:51, effects_chain.c:1124-1190).  `dsp -p` needs no device: the unmodified reference CLI linked
against libdsp_amd.so (oracle/_ref/dsp_gpu) prints a gnuplot script from this library's `plot` callbacks, the stock CLI
(oracle/_ref/dsp_ref) prints the reference's.  Both scripts are parsed into Python functions and the channel transfer functions
Ht<k>(f) are compared at a grid of frequencies; where the reference prints stored numbers (biquad, gain, direct FIR, remix,
st2ms, crossfeed, integer delay) the text itself must be identical.  The reference's FFT forms of fir / fir_p print their
taps after a transform round trip (fir.c:163-178, fir_p.c:209-233): same number of terms, values to 1e-15 of the peak."""
import cmath
import math
import os
import re
import subprocess

import numpy as np
import pytest

from oracle_api import REF_DIR

REF = os.path.join(REF_DIR, "dsp_ref")
GPU = os.path.join(REF_DIR, "dsp_gpu")
pytestmark = pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(GPU)), reason="oracle/_ref/dsp_gpu or dsp_ref not built")

COEFS40 = "coefs:" + ",".join(f"{0.9 ** i * (1 if i % 3 else -1):.6f}" for i in range(40))
COEFS12 = "coefs:" + ",".join(f"{(-0.7) ** i:.8f}" for i in range(12))
COEFS2CH = "coefs:" + ",".join(f"{0.8 ** i:.6f}" for i in range(50)) + "/" + ",".join(f"{(-0.6) ** i:.6f}" for i in range(37))


def script(exe, channels, chain, fs="48k"):
    r = subprocess.run([exe, "-p", "-r", fs, "-c", str(channels), "-n"] + chain.split(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    return r.returncode, r.stdout, r.stderr


DEF = re.compile(r"^([A-Za-z_][A-Za-z0-9_]*)\((\w)\)=(.*)$")


class Plot:
    """the function definitions of one gnuplot script, evaluated in Python"""

    def __init__(self, text):
        self.defs = {}
        for line in text.splitlines():
            m = DEF.match(line)
            if m:
                self.defs[m.group(1)] = (m.group(2), self._py(m.group(3)))
        self.env = {"exp": cmath.exp, "abs": abs, "pi": math.pi, "j": 1j, "log10": math.log10, "arg": cmath.phase, "complex": complex}
        for name in self.defs:
            self.env[name] = (lambda nm: (lambda x: self.call(nm, x)))(name)

    @staticmethod
    def _py(expr):
        expr = expr.replace("(abs(w)<=pi)?", "").replace(":0/0", "")
        return re.sub(r"\{([^{},]+),([^{},]+)\}", r"complex(\1,\2)", expr)

    def call(self, name, x):
        arg, expr = self.defs[name]
        return eval(expr, dict(self.env, **{arg: x}))            # noqa: S307 (test-only: both scripts come from our own binaries)


def compare_scripts(channels_out, chain, channels_in=None, fs="48k", tol=1e-9, same_text=False):
    ci = channels_in or channels_out
    rc_r, ref, err_r = script(REF, ci, chain, fs)
    rc_g, gpu, err_g = script(GPU, ci, chain, fs)
    assert rc_r == rc_g == 0, (chain, err_r[-400:], err_g[-400:])
    assert ("plot Ht0_mag_dB" in ref) and ("plot Ht0_mag_dB" in gpu), (chain, err_r[-300:], err_g[-300:])
    lr, lg = ref.splitlines(), gpu.splitlines()
    assert len(lr) == len(lg), chain
    if same_text:
        assert ref == gpu, chain
    # same definitions in the same order (names and argument letters: `(f)=1.0` of the no-op plot included)
    assert [l.split("=")[0] for l in lr] == [l.split("=")[0] for l in lg], chain
    # the FIR forms: same number of terms per channel
    assert [l.count("+exp(-j*w*") for l in lr] == [l.count("+exp(-j*w*") for l in lg], chain
    pr, pg = Plot(ref), Plot(gpu)
    fsn = pr_rate(ref)
    worst = 0.0
    for k in range(channels_out):
        for f in (11.0, 97.3, 440.0, 1234.5, 5000.0, 0.23 * fsn, 0.41 * fsn, 0.499 * fsn):
            a, b = pr.call(f"Ht{k}", f), pg.call(f"Ht{k}", f)
            worst = max(worst, abs(a - b) / max(abs(a), 1e-3))
    assert worst <= tol, (chain, worst)


def pr_rate(text):
    return float(re.search(r"set xrange \[10:(\d+)/2\]", text).group(1))


@pytest.mark.parametrize("channels,chain", [
    (2, "gain -6 lowpass 1k 0.707 eq 400 2.0 1.5 :1 highshelf 8k 6d -3"),
    (2, f"fir {COEFS12} :0 fir -a {COEFS12}"),                                      # direct form (<= 16 taps), zero-padded to 16 terms
    (3, "remix 0,1 2 . :0,2 st2ms"),
    (4, ":1,3 ms2st :0,2 crossfeed 700 4.5 :2 add 0.01 : mult 0.5"),
    (2, "delay 11S :1 delay -3S"),
    (2, f"fir_p coefs:1,0.5,0.25,0.125 remix 1 0"),                                 # fir_p of <= 32 taps: the direct form again
])
def test_plot_text_identical(channels, chain):
    compare_scripts(channels, chain, same_text=True)


def random_section(rng):
    """one effect of the biquad family with random arguments in the reference's own notations (README.md:169-236)"""
    def f0(lo=20.0, hi=20000.0):
        f = float(np.exp(rng.uniform(np.log(lo), np.log(hi))))
        return f"{f / 1000.0:.4g}k" if rng.random() < 0.4 else f"{f:.5g}"
    def width(shelf=False):
        kinds = ["q", "", "o", "h", "k"] + (["s", "d"] if shelf else [])
        k = kinds[int(rng.integers(len(kinds)))]
        if k in ("q", ""):
            return f"{rng.uniform(0.3, 6.0):.4g}{k}"
        if k == "o":
            return f"{rng.uniform(0.2, 3.0):.4g}o"
        if k == "h":
            return f"{rng.uniform(5.0, 400.0):.4g}h"
        if k == "k":
            return f"{rng.uniform(0.01, 0.8):.4g}k"
        if k == "s":
            return f"{rng.uniform(0.3, 1.0):.4g}s"
        return f"{rng.uniform(3.0, 12.0):.4g}d"
    gain = lambda: f"{rng.uniform(-12.0, 12.0):+.3g}"
    name = ["lowpass_1", "highpass_1", "allpass_1", "lowshelf_1", "highshelf_1", "lowpass_1p", "lowpass", "highpass", "bandpass_skirt",
            "bandpass_peak", "notch", "allpass", "eq", "lowshelf", "highshelf", "lowpass_transform", "highpass_transform",
            "linkwitz_transform", "deemph", "biquad", "bw"][int(rng.integers(21))]
    if name in ("lowpass_1", "highpass_1", "allpass_1", "lowpass_1p"):
        return f"{name} {f0()}"
    if name in ("lowshelf_1", "highshelf_1"):
        return f"{name} {f0()} {gain()}"
    if name in ("lowpass", "highpass", "bandpass_skirt", "bandpass_peak", "notch", "allpass"):
        return f"{name} {f0()} {width()}"
    if name == "eq":
        return f"eq {f0()} {width()} {gain()}"
    if name in ("lowshelf", "highshelf"):
        return f"{name} {f0()} {width(True)} {gain()}"
    if name in ("lowpass_transform", "highpass_transform", "linkwitz_transform"):
        return f"{name} {f0(20, 200)} {rng.uniform(0.4, 1.5):.3g} {f0(15, 150)} {rng.uniform(0.4, 1.0):.3g}"
    if name == "deemph":
        return "deemph"
    if name == "biquad":
        c = rng.uniform(-0.9, 0.9, size=6)
        return "biquad " + " ".join(f"{v:.6g}" for v in (c[0], c[1], c[2], 1.0 + 0.2 * c[3], 0.5 * c[4], 0.2 * c[5]))
    order = int(rng.integers(2, 9))                                      # Butterworth / Linkwitz-Riley sections: bw<order>[.<index>]
    return f"{['lowpass', 'highpass'][int(rng.integers(2))]} {f0(40, 8000)} bw{order}" + (f".{int(rng.integers(order // 2 + order % 2))}" if rng.random() < 0.7 else "")


@pytest.mark.parametrize("seed", range(24))
def test_random_biquad_family_arguments_print_the_reference_text(seed):
    """every effect of the biquad family with random arguments in every notation of the reference (frequencies with k, widths as
    q / o / h / k / s / d, bw<n>.<k> sections, gains): `dsp -p` through this library prints the reference's coefficients digit for digit"""
    rng = np.random.default_rng(4000 + seed)
    channels = int(rng.integers(1, 4))
    parts = []
    for _ in range(int(rng.integers(3, 9))):
        if channels > 1 and rng.random() < 0.3:
            parts.append(":" + ",".join(str(c) for c in sorted(rng.choice(channels, size=int(rng.integers(1, channels + 1)), replace=False))))
        if rng.random() < 0.2:
            parts.append(f"gain {rng.uniform(-9, 3):.3g}")
        parts.append(random_section(rng))
    fs = ["44.1k", "48k", "96k"][int(rng.integers(3))]
    chain = " ".join(parts)
    rc_r, _, err_r = script(REF, channels, chain, fs)
    if rc_r != 0:
        # (an argument the reference refuses -- a bw<n>.<k> index beyond the order, a frequency beyond fs / 2: the same verdict in the same words)
        rc_g, _, err_g = script(GPU, channels, chain, fs)
        assert rc_g == rc_r, (chain, err_r[-300:], err_g[-300:])
        # (messages carry the host's program name -- the library reads the host's dsp_globals: only the two binaries' names differ)
        norm = lambda t: t.replace("dsp_ref", "dsp_X").replace("dsp_gpu", "dsp_X")
        assert norm(err_g) == norm(err_r), (chain, err_r[-300:], err_g[-300:])
        return
    compare_scripts(channels, chain, fs=fs, same_text=True)


@pytest.mark.parametrize("seed", range(24))
def test_random_chains_print_the_reference_text(seed):
    """whole chains of the effects whose plots are stored numbers -- sections, gain / mult / add (merged by the host), remix (channel
    counts change), integer delays, st2ms / ms2st, crossfeed -- under random selectors: the same gnuplot script, character for character"""
    rng = np.random.default_rng(9000 + seed)
    channels = int(rng.integers(1, 6))
    ch, parts = channels, []

    def selector():
        k = int(rng.integers(1, ch + 1))
        return ":" + ",".join(str(c) for c in sorted(rng.choice(ch, size=k, replace=False)))

    for _ in range(int(rng.integers(3, 10))):
        r = rng.random()
        if ch > 1 and r < 0.25:
            parts.append(selector())
            continue
        if r < 0.45:
            parts.append(random_section(rng))
        elif r < 0.6:
            parts.append(str(rng.choice(["gain", "mult", "add"])) + f" {rng.uniform(-6, 6):.3g}")
        elif r < 0.7:
            parts.append(f"delay {int(rng.integers(0, 200))}S" if rng.random() < 0.5 else f"delay {rng.uniform(0, 3):.3g}m")
        elif r < 0.8 and ch >= 2:
            pair = sorted(rng.choice(ch, size=2, replace=False))
            parts.append(f":{pair[0]},{pair[1]}")
            parts.append(str(rng.choice(["st2ms", "ms2st"])) if rng.random() < 0.6 else f"crossfeed {rng.uniform(300, 1200):.4g} {rng.uniform(2, 9):.3g}")
            parts.append(":")
        elif r < 0.9:
            parts.append(":")                                                           # remix acts on its own argument list, not the selector
            new_ch = int(rng.integers(1, 5))
            outs = []
            for _o in range(new_ch):
                k = int(rng.integers(0, min(ch, 3) + 1))
                outs.append(",".join(str(c) for c in sorted(rng.choice(ch, size=k, replace=False))) if k else ".")
            parts.append("remix " + " ".join(outs))
            ch = new_ch
        else:
            parts.append(":")
    chain = " ".join(parts)
    rc_r, _, err_r = script(REF, channels, chain)
    if rc_r != 0:
        rc_g, _, err_g = script(GPU, channels, chain)
        assert rc_g == rc_r and err_g.replace("dsp_gpu", "dsp_X") == err_r.replace("dsp_ref", "dsp_X"), (chain, err_r[-300:], err_g[-300:])
        return
    compare_scripts(ch, chain, channels_in=channels, same_text=True)


@pytest.mark.parametrize("channels,chain,cin", [
    (2, f"fir {COEFS40}", None),                                                      # FFT form: next_fast_fftw_len(40) terms
    (2, f"fir_p {COEFS40} :0 fir_p -a {COEFS40}", None),                              # 32 direct + one 32-tap partition
    (2, f"fir_p {COEFS2CH}", None),                                                   # one filter per channel, ragged
    (2, "hilbert 127 :1 hilbert -p -a 45 255", None),
    (3, f"gain -3 remix 0 1,2 0,2 fir {COEFS40} :1,2 st2ms", 3),
    (2, "delay -f 0.3S :1 delay -f1 2.7S", None),
    (2, "delay -f5 2.3S :0 delay -f3 0.4S", None),                                    # the reference prints the ladder, this library its sections
    (2, "delay -f34 0.3S :1 delay -f41 12.77S", None),                                # round 4: the poles of orders above 33 in 113-bit arithmetic (thiran_roots.cpp)
    (1, "delay -f50 0.5S", None),
    (2, "lowpass -r 1k 0.707", None),
    (2, "lowpass 2k 0.707 lowpass -r 2k 0.707 :0 highpass -r60 30 0.707", None),
    (1, "eq -r 400 2.0 1.5 highshelf -r 8k 0.7 -3 lowpass_1 -r 300", None),
    (2, "lowpass -r 2k 0.707 lowpass -r 2k 0.707", None),                             # repeated poles: series states
])
def test_plot_same_function(channels, chain, cin):
    compare_scripts(channels, chain, channels_in=cin, tol=2e-8)


def test_plot_long_partitioned_filter_term_counts(tmp_path):
    # the number of terms fir_p prints follows its partition plan (fir_p.c:242-289): 32 + groups of zero-padded partitions
    rng = np.random.Generator(np.random.PCG64(5))
    for T, extra in ((100, ""), (1000, ""), (4095, ""), (4096, ""), (5000, "64"), (20000, ""), (70001, ""), (70001, "1024"), (140000, "32768")):
        h = rng.standard_normal(T) * np.exp(-np.arange(T) / (T / 8.0))
        f = os.path.join(str(tmp_path), f"h{T}.raw"); h.astype("<f8").tofile(f)
        chain = f"fir_p -t pcm -e double -c 1 {extra} {f}"
        rc_r, ref, _ = script(REF, 1, chain)
        rc_g, gpu, _ = script(GPU, 1, chain)
        assert rc_r == rc_g == 0
        lr = [l for l in ref.splitlines() if l.startswith("H0_0(w)=")][0]
        lg = [l for l in gpu.splitlines() if l.startswith("H0_0(w)=")][0]
        nr, ng = lr.count("+exp(-j*w*"), lg.count("+exp(-j*w*")
        assert nr == ng, (T, extra, nr, ng)
        vr = np.array([float(t.split("*")[-1]) for t in lr[lr.index("(0.0") + 4:lr.rindex("):0/0")].split("+exp")[1:]])
        vg = np.array([float(t.split("*")[-1]) for t in lg[lg.index("(0.0") + 4:lg.rindex("):0/0")].split("+exp")[1:]])
        assert np.max(np.abs(vr - vg)) <= 1e-14 * np.max(np.abs(vr)), (T, extra)


def test_effects_without_plot_are_refused_alike():
    # resample and zita_convolver have no plot callback in the reference either: same verdict from plot_effects_chain
    for chain in ("resample 96k", "gain -3 resample 44.1k"):
        rc_r, ref, err_r = script(REF, 2, chain)
        rc_g, gpu, err_g = script(GPU, 2, chain)
        assert rc_r == rc_g
        assert ("does not support plotting" in err_r) == ("does not support plotting" in err_g), (err_r, err_g)
        assert ref == gpu


def test_messages_follow_the_hosts_program_name_and_verbosity():
    """the library's log lines carry the HOST's program name (dsp_globals.prog_name, dsp.h:44-47) and obey its -v / -q: a refused
    argument reads the same through both builds, -q silences the library too, -v lets its info lines through"""
    bad = ["-p", "-r", "48k", "-c", "1", "-n", "lowpass", "1k", "bw5.7"]
    r = subprocess.run([REF] + bad, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    g = subprocess.run([GPU] + bad, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == g.returncode != 0
    assert g.stderr.replace("dsp_gpu", "dsp_X") == r.stderr.replace("dsp_ref", "dsp_X"), (r.stderr, g.stderr)
    assert GPU in g.stderr                                                       # argv[0] as the reference prints it, not a fixed "dsp"
    q = subprocess.run([GPU, "-q"] + bad, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    qr = subprocess.run([REF, "-q"] + bad, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert q.returncode != 0 and q.stderr.replace("dsp_gpu", "dsp_X") == qr.stderr.replace("dsp_ref", "dsp_X"), (q.stderr, qr.stderr)


ODD_ARGUMENTS = [
    (2, "lowpass 30k 0.7"), (2, "lowpass 1k 0"), (2, "lowpass -r5 1k 0.7"), (2, "lowpass -r300 1k 0.7"), (2, "lowpass -rx 1k 0.7"), (2, "lowpass 1kk 0.7"), (2, "lowpass 1k 0.7z"),
    (2, "eq 1k 0q 3"), (2, "eq 1k 1q x"), (2, "lowshelf 1k -1s 3"), (2, "highshelf 1k 0d 3"), (2, "lowpass 1k bw0"), (2, "lowpass 1k bw3.5"), (2, "lowpass 1k bwx"),
    (2, "lowpass_1 -1"), (2, "linkwitz_transform 80 0 40 0.5"), (2, "linkwitz_transform 80 0.9 40000 0.5"), (2, "biquad 1 2 3 x 1 1"), (2, "lowpass"), (2, "eq 1k 1q 3 4"),
    (2, "crossfeed 30k 3"), (2, "crossfeed 700 -1"), (2, "crossfeed 700 x"), (3, "crossfeed 700 3"), (1, "crossfeed 700 3"), (2, "crossfeed"),
    (1, "st2ms"), (3, "st2ms"), (3, "ms2st"), (2, "st2ms 1"),
    (2, "delay -f0 1S"), (2, "delay -f51 1S"), (2, "delay 1x"), (2, "delay"), (2, "delay -fx 1S"), (2, "delay -z 1S"),
    (2, "resample 0"), (2, "resample -b 0.5 44.1k"), (2, "resample -b 1 44.1k"), (2, "resample -bx 44.1k"), (2, "resample 44.1kk"), (2, "resample"), (2, "resample -1"), (2, "resample x"),
    (2, "remix 5"), (2, "remix x"), (2, "remix"), (2, "remix 0-"), (2, "remix 1-0"), (2, "remix 0,,1"),
    (2, "gain x"), (2, "gain"), (2, "mult"), (2, "add"), (2, "gain 1 2"), (2, "mult 1x"),
    (2, "fir coefs:"), (2, "fir coefs:1,x"), (2, "fir coefs:1,2/3/4"), (2, "fir_p x coefs:1"), (2, "fir -ax coefs:1,2"), (2, "fir"), (2, "fir_p 1 2 coefs:1"),
    (2, "fir -t pcm nofile"), (2, "fir_p -e s16 -t pcm nofile"), (2, "fir nofile"), (2, "fir -z coefs:1"),
    (2, "fir_p 48 coefs:" + ",".join(["0.1"] * 40)), (2, "fir_p 16 coefs:" + ",".join(["0.1"] * 40)),        # max_part_len: not a power of two / below 32 (filters of more than 32 taps)
    (2, "hilbert 0"), (2, "hilbert 10"), (2, "hilbert -a x 127"), (2, "hilbert"), (2, "hilbert 127 3"),
    (2, ":5 gain 1"), (2, ":x gain 1"), (2, ":, gain 1"),
]
ACCEPTED_ALIKE = [(2, "fir_p 3 coefs:1,2"), (2, "fir_p 16 8 coefs:1,2"), (2, "fir_p 0 coefs:1"), (2, "remix 0,1 ."), (2, "biquad 1 2 3 0 1 1"), (2, "hilbert -p 127"), (2, "fir -a coefs:1,2")]


@pytest.mark.parametrize("channels,chain", ODD_ARGUMENTS + ACCEPTED_ALIKE)
def test_refusals_read_the_same_through_both_builds(channels, chain):
    """arguments the reference refuses, most of them (and a few it accepts against expectation: a `max_part_len` in front of a filter of up to 32 taps is
    never looked at, fir_p.c:364-384): same exit status, and on stderr the same lines -- the parsers' own notes ("parse_freq: error:
    trailing characters"), `parameter out of range: <what>` (util.c:541-563), the usage line where the reference prints it"""
    args = ["-p", "-r", "48k", "-c", str(channels), "-n"] + chain.split()
    r = subprocess.run([REF] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    g = subprocess.run([GPU] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == g.returncode, (chain, r.stderr[-300:], g.stderr[-300:])
    assert g.stderr.replace("dsp_gpu", "dsp_X") == r.stderr.replace("dsp_ref", "dsp_X"), (chain, r.stderr[-400:], g.stderr[-400:])
