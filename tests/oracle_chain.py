"""A tiny chain interpreter over the CPU oracle (tests only).

Interprets the subset of the reference's chain language used by the golden
cases (effect names + args, ':selector' tokens; effects_chain.c:445-603) and
evaluates each effect with oracle/dsp_oracle.c, with the CLI's whole-stream
semantics: drain taps-1 frames for FIRs (fir_p.c:235-240), discard fir's
latency at the end of the chain (align.c:147-152), realise integer delays with
the align ring (delay.c:142-147, align.c:35-44), flush resample with drain2
(resample.c:163-188).
"""
import ctypes as C

import numpy as np

from oracle_api import Oracle as O

BIQ = {"lowpass_1": (1, 1), "highpass_1": (2, 1), "allpass_1": (3, 1), "lowshelf_1": (4, 2), "highshelf_1": (5, 2),
       "lowpass_1p": (6, 1), "lowpass": (7, 2), "highpass": (8, 2), "bandpass_skirt": (9, 2), "bandpass_peak": (10, 2),
       "notch": (11, 2), "allpass": (12, 2), "eq": (13, 3), "lowshelf": (14, 3), "highshelf": (15, 3),
       "lowpass_transform": (16, 4), "highpass_transform": (17, 4), "linkwitz_transform": (17, 4),
       "deemph": (18, 0), "biquad": (19, 6)}
OTHER = {"gain": 1, "mult": 1, "add": 1, "remix": None, "delay": 1, "fir": None, "fir_p": None, "resample": None, "hilbert": None,
         "st2ms": 0, "ms2st": 0, "crossfeed": 2}


def parse_freq(s):
    return float(s[:-1]) * 1000.0 if s.endswith("k") else float(s)


def parse_selector(s, n):
    sel = np.zeros(n, dtype=bool)
    if s in ("", "-"):
        sel[:] = True
        return sel
    for part in s.split(","):
        if "-" in part:
            a, b = part.split("-")
            a = int(a) if a else 0
            b = int(b) if b else n - 1
            sel[a:b + 1] = True
        else:
            sel[int(part)] = True
    return sel


def tokenize(chain):
    toks = chain.split()
    out = []
    i = 0
    while i < len(toks):
        t = toks[i]
        if t.startswith(":"):
            out.append((":", [t[1:]]))
            i += 1
            continue
        assert t in BIQ or t in OTHER, t
        j = i + 1
        while j < len(toks) and not (toks[j] in BIQ or toks[j] in OTHER or toks[j].startswith(":")):
            j += 1
        out.append((t, toks[i + 1:j]))
        i = j
    return out


def biquad_coefs(name, args, fs):
    t, _ = BIQ[name]
    if name == "biquad":
        return O.biquad_coefs(*[float(a) for a in args])
    if name == "deemph":
        p = {44100: (5283.0, 0.4845, -9.477), 48000: (5356.0, 0.479, -9.62)}[fs]
        return O.biquad_design(15, fs, p[0], p[1], p[2], 0, 2)
    if t in (1, 2, 3, 6):
        return O.biquad_design(t, fs, parse_freq(args[0]))
    if t in (4, 5):
        return O.biquad_design(t, fs, parse_freq(args[0]), 0, float(args[1]))
    if t in (16, 17):
        return O.biquad_design(t, fs, parse_freq(args[0]), float(args[1]), parse_freq(args[2]), float(args[3]))
    w, wt, ok = O.parse_width(args[1])
    assert ok
    g = float(args[2]) if len(args) > 2 else 0.0
    return O.biquad_design(t, fs, parse_freq(args[0]), w, g, 0, wt)


def load_filter(args, filt):
    """returns taps [T, fch]"""
    spec = args[-1]
    if spec.startswith("coefs:"):
        chans = [[float(v) for v in c.split(",")] for c in spec[6:].split("/")]
        T = max(len(c) for c in chans)
        h = np.zeros((T, len(chans)))
        for k, c in enumerate(chans):
            h[:len(c), k] = c
        return h
    h = np.asarray(filt, dtype=np.float64)
    return h[:, None] if h.ndim == 1 else h


def build(chain, fs, ch, filt=None):
    """Parse into effect records (what each effect's init() would hold)."""
    effs = []
    sel = np.ones(ch, dtype=bool)
    for name, args in tokenize(chain):
        if name == ":":
            sel = parse_selector(args[0], ch)
            continue
        if len(sel) != ch:
            sel = np.ones(ch, dtype=bool)
        e = dict(name=name, ifs=fs, ofs=fs, ich=ch, och=ch, sel=sel.copy(), merge=None, reorder=False)
        if name in BIQ:
            c = biquad_coefs(name, args, fs)
            e.update(kind="biquad", coefs={int(k): c for k in np.nonzero(sel)[0]}, merge="biquad", reorder=True)
        elif name in ("gain", "mult", "add"):
            v = float(args[0])
            if name == "gain":
                v = 10.0 ** (v / 20.0)
            noop = 0.0 if name == "add" else 1.0
            e.update(kind="add" if name == "add" else "gain", vec=np.where(sel, v, noop).astype(np.float64),
                     merge="add" if name == "add" else "gain", reorder=(name != "add"))
        elif name == "remix":
            nsel = len(args)
            och = ch + nsel - int(sel.sum())
            m = np.zeros((och, ch), dtype=np.int8)
            i = 0
            c_in = 0
            for k in range(och):
                if c_in >= ch or sel[c_in]:
                    if i < nsel:
                        if args[i] != ".":
                            sub = parse_selector(args[i], int(sel.sum()))
                            m[k, np.nonzero(sel)[0][sub]] = 1
                        i += 1
                    else:
                        while c_in < ch and sel[c_in]:
                            c_in += 1
                        if c_in < ch:
                            m[k, c_in] = 1
                else:
                    m[k, c_in] = 1
                c_in += 1
            e.update(kind="remix", mat=m, och=och)
            ch = och
        elif name == "delay":
            # delay.c:688-742: [-f[order]] amount[s|m|S]; without -f the amount is rounded to whole samples (lrint)
            a = list(args)
            frac, order = False, 0
            if a[0].startswith("-f"):
                frac, order = True, (int(a[0][2:]) if len(a[0]) > 2 else 0)
                a = a[1:]
            t = a[0]
            v = float(t[:-1]) if t.endswith("S") else float(t[:-1]) * fs / 1000.0 if t.endswith("m") else float(t.rstrip("s")) * fs
            e.update(kind="delay", n=np.where(sel, 0 if frac else int(np.rint(v)), 0), frac=np.where(sel, v if frac else 0.0, 0.0),
                     apn=np.where(sel, order, 0), merge="delay", reorder=True)
        elif name in ("fir", "fir_p", "hilbert"):
            if name == "hilbert":
                h = O.hilbert_taps(int(args[-1]))[:, None]
                kind = "fir_p" if "-p" in args else "fir"
            else:
                h = load_filter(args, filt)
                kind = name
            T = h.shape[0]
            if (kind == "fir_p" and T <= 32) or (kind == "fir" and T <= 16):
                kind = "fir_direct"
            e.update(kind=kind, taps=h, reorder=True)
        elif name == "resample":
            rate = int(round(parse_freq(args[-1])))
            e.update(kind="resample", ofs=rate)
            fs = rate
        elif name in ("st2ms", "ms2st"):
            # st2ms.c:28-54: exactly two selected channels (st2ms.c:96-99), sum / difference, halved for st2ms
            pair = np.nonzero(sel)[0]
            assert len(pair) == 2, "st2ms / ms2st need two selected channels"
            e.update(kind="midside", pair=(int(pair[0]), int(pair[1])), scale=0.5 if name == "st2ms" else None)
        elif name == "crossfeed":
            # crossfeed.c:33-50, 128-137: direct + low-passed opposite + high-passed own channel (first-order sections at f0)
            pair = np.nonzero(sel)[0]
            assert len(pair) == 2, "crossfeed needs two selected channels"
            sep = 10.0 ** (float(args[1]) / 20.0)
            e.update(kind="crossfeed", pair=(int(pair[0]), int(pair[1])), direct=sep / (1 + sep), cross=1 / (1 + sep),
                     lp=biquad_coefs("lowpass_1", [args[0]], fs), hp=biquad_coefs("highpass_1", [args[0]], fs))
        else:
            raise ValueError(name)
        effs.append(e)
    return effs


def optimize(effs):
    """effects_chain.c:605-641: forward scan from each merge-capable effect; a
    candidate without merge() is skipped only if reorderable, a refused merge
    is skipped too (so gains merge across biquads -- and across `add`)."""
    i = 0
    while i < len(effs):
        d = effs[i]
        if d["merge"]:
            j = i + 1
            while j < len(effs):
                s = effs[j]
                if (s["ifs"], s["ich"], s["ofs"], s["och"]) != (d["ifs"], d["ich"], d["ofs"], d["och"]):
                    break
                if s["merge"] is None:
                    if s["reorder"]:
                        j += 1
                        continue
                    break
                merged = False
                if s["merge"] == d["merge"]:
                    if d["merge"] == "gain":
                        d["vec"] = d["vec"] * s["vec"]; merged = True
                    elif d["merge"] == "add":
                        d["vec"] = d["vec"] + s["vec"]; merged = True
                    elif d["merge"] == "delay":      # delay.c:126-140
                        d["n"] = d["n"] + s["n"]; d["frac"] = d["frac"] + s["frac"]; d["apn"] = np.maximum(d["apn"], s["apn"]); merged = True
                    elif d["merge"] == "biquad" and not (set(d["coefs"]) & set(s["coefs"])):
                        d["coefs"].update(s["coefs"]); merged = True   # biquad.c:344-376
                if merged:
                    del effs[j]
                else:
                    j += 1
        i += 1
    return effs


def prepare(effs):
    """delay.c:149-205: the fractional part of a (merged) delay goes to a Thiran all-pass of order apn (default 2), which by itself
    delays by apn - 1 + fraction samples; the integer request is reduced by as much (it may become negative: the host's alignment
    then treats the channel as late, align.c:125-152)."""
    for e in effs:
        if e["kind"] != "delay":
            continue
        n, fr, apn = e["n"].astype(np.int64), e["frac"].astype(np.float64), np.where(e["apn"] < 1, 2, e["apn"]).astype(np.int64)
        for k in range(len(n)):
            if abs(fr[k] - np.rint(fr[k])) >= np.finfo(np.float64).eps:
                adj = int(apn[k] - 1) - int(np.floor(fr[k] - 0.1))
                n[k] -= adj
                fr[k] += adj
            else:
                n[k] += int(np.rint(fr[k]))
                fr[k] = 0.0
                apn[k] = 0
        e.update(n=n, frac=fr, apn=apn)
    return effs


def drain_frames(effs):
    """effects_chain.c:877-923 for identity-dependency chains (+ remix via its matrix)."""
    if not effs:
        return 0
    samples = np.zeros(effs[0]["ich"], dtype=np.int64)
    for e in effs:
        k = e["kind"]
        if k == "remix":
            m = e["mat"].astype(bool)
            samples = np.array([samples[m[i]].max() if m[i].any() else 0 for i in range(e["och"])], dtype=np.int64)
        elif k in ("fir_p", "fir_direct"):
            samples = samples + np.where(e["sel"], e["taps"].shape[0] - 1, 0)
        elif k == "fir":
            T = e["taps"].shape[0]
            samples = samples + np.where(e["sel"], int(O.lib().orc_next_fast_fftw_len(T)) + T - 1, 0)
        elif k == "delay":
            n = e["n"] - min(0, int(e["n"].min()))
            samples = samples + n + e["apn"]             # delay.c:105-110
        elif k == "resample":
            g = np.gcd(e["ofs"], e["ifs"])
            n, d = e["ofs"] // g, e["ifs"] // g
            samples = -(-samples * n // d)
    D = int(samples.max()) if len(samples) else 0
    ifs, ofs = effs[0]["ifs"], effs[-1]["ofs"]
    if ifs != ofs:
        g = np.gcd(ifs, ofs)
        D = D * (ifs // g) // (ofs // g)
    return D


def run(chain, x, fs, filt=None):
    """Whole-stream evaluation with the CLI's semantics; x [frames, ch] -> (y [oframes, och], ofs).

    The host pushes chain.drain_frames zero frames through the WHOLE chain after
    the input ends (effects_chain.c:1193-1198), so IIR tails ring into later FIRs."""
    L = O.lib()
    x = np.ascontiguousarray(x, dtype=np.float64).copy()
    effs = prepare(optimize(build(chain, fs, x.shape[1], filt)))
    x = np.vstack([x, np.zeros((drain_frames(effs), x.shape[1]))])
    discard = 0
    for e in effs:
        ch = x.shape[1]
        kind = e["kind"]
        if kind == "biquad":
            for k, c in e["coefs"].items():
                m = np.zeros(2)
                L.orc_biquad_run(c.ctypes.data, m.ctypes.data, x.ctypes.data + 8 * int(k), x.shape[0], ch)
        elif kind in ("gain", "add"):
            vec = np.ascontiguousarray(e["vec"])
            (L.orc_add_run if kind == "add" else L.orc_gain_run)(x.ctypes.data, x.shape[0], ch, vec.ctypes.data)
        elif kind == "remix":
            out = np.zeros((x.shape[0], e["och"]))
            m = np.ascontiguousarray(e["mat"])
            L.orc_remix_run(x.ctypes.data, out.ctypes.data, x.shape[0], ch, e["och"], m.ctypes.data)
            x = out
        elif kind == "delay":
            for k in range(ch):
                if e["apn"][k] > 0:
                    st = np.zeros(max(4, int(e["apn"][k])))
                    L.orc_frac_delay_run(x.ctypes.data + 8 * k, x.shape[0], ch, int(e["apn"][k]), abs(float(e["frac"][k])), st.ctypes.data)
            # the integer part is realised by an align effect with the requested lengths; requests that went negative (the
            # all-pass's own delay) count from the most negative one (align.c:125-146)
            n = e["n"] - min(0, int(e["n"].min()))
            for k in range(ch):
                if n[k] > 0:
                    ring = np.zeros(int(n[k]))
                    p = C.c_ssize_t(0)
                    L.orc_delay_run(x.ctypes.data + 8 * k, x.shape[0], ch, ring.ctypes.data, int(n[k]), C.byref(p))
        elif kind in ("fir", "fir_p", "fir_direct"):
            h = e["taps"]
            T = h.shape[0]
            idx = np.nonzero(e["sel"])[0]
            for j, k in enumerate(idx):
                t = np.ascontiguousarray(h[:, j if h.shape[1] > 1 else 0])
                new = getattr(L, f"orc_{kind}_new")
                st = new(t.ctypes.data, T, 0) if kind == "fir_p" else new(t.ctypes.data, T)
                getattr(L, f"orc_{kind}_run")(st, x.ctypes.data + 8 * int(k), x.shape[0], ch)
                getattr(L, f"orc_{kind}_free")(st)
            if kind == "fir":
                # every selected channel is late by len; the host drops the common latency (align.c:147-152)
                assert len(idx) == ch, "oracle_chain: fir on a channel subset not modelled"
                x = x[int(L.orc_next_fast_fftw_len(T)):]
        elif kind == "resample":
            x = O.resample(x, e["ifs"], e["ofs"])
            fs = e["ofs"]
        elif kind == "midside":
            c0, c1 = e["pair"]
            s0, s1 = x[:, c0].copy(), x[:, c1].copy()
            x[:, c0] = (s0 + s1) if e["scale"] is None else (s0 + s1) * e["scale"]      # every operation rounds once, as in C
            x[:, c1] = (s0 - s1) if e["scale"] is None else (s0 - s1) * e["scale"]
        elif kind == "crossfeed":
            c0, c1 = e["pair"]
            s0, s1 = x[:, c0].copy(), x[:, c1].copy()

            def section(c, v):
                y, m = np.ascontiguousarray(v.copy()), np.zeros(2)
                L.orc_biquad_run(c.ctypes.data, m.ctypes.data, y.ctypes.data, y.shape[0], 1)
                return y
            x[:, c0] = (s0 * e["direct"]) + (section(e["lp"], s1) * e["cross"]) + (section(e["hp"], s0) * e["cross"])
            x[:, c1] = (s1 * e["direct"]) + (section(e["lp"], s0) * e["cross"]) + (section(e["hp"], s1) * e["cross"])
    return x[discard:], fs
