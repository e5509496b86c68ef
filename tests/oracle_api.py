"""ctypes bindings for the TEST-ONLY checkers.

* ``Oracle``  -> oracle/liboracle.so  (CPU restatement, oracle/dsp_oracle.c)
* ``RefChain`` -> oracle/_ref/libdspref*.so (the real reference, built by
  oracle/Makefile from /root/reference; prebuilt files travel to the GPU box)

Nothing under dsp_amd/ imports this module.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")

_dp = C.POINTER(C.c_double)
_ss = C.c_ssize_t


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Thin wrapper over oracle/dsp_oracle.h; arrays are numpy float64."""

    _lib = None

    @classmethod
    def available(cls):
        return os.path.exists(ORACLE_SO)

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(ORACLE_SO)
            L.orc_next_fast_fftw_len.restype = _ss
            L.orc_next_fast_fftw_len.argtypes = [_ss]
            L.orc_parse_width.restype = C.c_double
            L.orc_parse_width.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
            L.orc_biquad_coefs.argtypes = [C.c_double] * 6 + [C.c_void_p]
            L.orc_biquad_design.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]
            L.orc_biquad_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, _ss, C.c_int]
            L.orc_gain_run.argtypes = [C.c_void_p, _ss, C.c_int, C.c_void_p]
            L.orc_add_run.argtypes = [C.c_void_p, _ss, C.c_int, C.c_void_p]
            L.orc_remix_run.argtypes = [C.c_void_p, C.c_void_p, _ss, C.c_int, C.c_int, C.c_void_p]
            L.orc_delay_run.argtypes = [C.c_void_p, _ss, C.c_int, C.c_void_p, _ss, C.POINTER(_ss)]
            L.orc_frac_delay_run.argtypes = [C.c_void_p, _ss, C.c_int, C.c_int, C.c_double, C.c_void_p]
            for name in ("fir_direct", "fir"):
                getattr(L, f"orc_{name}_new").restype = C.c_void_p
                getattr(L, f"orc_{name}_new").argtypes = [C.c_void_p, _ss]
                getattr(L, f"orc_{name}_run").argtypes = [C.c_void_p, C.c_void_p, _ss, C.c_int]
                getattr(L, f"orc_{name}_free").argtypes = [C.c_void_p]
            L.orc_fir_latency.restype = _ss
            L.orc_fir_latency.argtypes = [C.c_void_p]
            L.orc_fir_p_plan.restype = C.c_int
            L.orc_fir_p_plan.argtypes = [_ss, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
            L.orc_fir_p_new.restype = C.c_void_p
            L.orc_fir_p_new.argtypes = [C.c_void_p, _ss, C.c_int]
            L.orc_fir_p_run.argtypes = [C.c_void_p, C.c_void_p, _ss, C.c_int]
            L.orc_fir_p_free.argtypes = [C.c_void_p]
            L.orc_resample_new.restype = C.c_void_p
            L.orc_resample_new.argtypes = [C.c_int, C.c_int, C.c_double]
            L.orc_resample_params.argtypes = [C.c_void_p, C.c_void_p]
            L.orc_resample_run.restype = _ss
            L.orc_resample_run.argtypes = [C.c_void_p, C.c_void_p, _ss, C.c_int, C.c_void_p, C.c_int]
            L.orc_resample_drain.restype = _ss
            L.orc_resample_drain.argtypes = [C.c_void_p, _ss, C.c_void_p, C.c_void_p]
            L.orc_resample_free.argtypes = [C.c_void_p]
            L.orc_hilbert_taps.argtypes = [_ss, C.c_double, C.c_void_p]
            L.orc_sgen_sine.argtypes = [C.c_void_p, _ss, C.c_int, C.c_int, C.c_double, _ss]
            L.orc_sgen_sweep.argtypes = [C.c_void_p, _ss, C.c_int, C.c_int, C.c_double, C.c_double, _ss, _ss]
            L.orc_sgen_delta.argtypes = [C.c_void_p, _ss, C.c_int, _ss, _ss]
            L.orc_conv_full.argtypes = [C.c_void_p, _ss, C.c_void_p, _ss, C.c_void_p]
            L.orc_zita_equiv_new.restype = C.c_void_p
            L.orc_zita_equiv_new.argtypes = [C.c_void_p, _ss, C.c_int]
            L.orc_zita_equiv_run.argtypes = [C.c_void_p, C.c_void_p, _ss, C.c_int]
            L.orc_zita_equiv_free.argtypes = [C.c_void_p]
            cls._lib = L
        return cls._lib

    # ---- convenience (whole-signal, [frames, channels] float64 arrays) ----
    @classmethod
    def biquad_design(cls, btype, fs, a0=0.0, a1=0.0, a2=0.0, a3=0.0, width_type=1):
        c = np.zeros(5)
        cls.lib().orc_biquad_design(btype, fs, a0, a1, a2, a3, width_type, _ptr(c))
        return c

    @classmethod
    def biquad_coefs(cls, b0, b1, b2, a0, a1, a2):
        c = np.zeros(5)
        cls.lib().orc_biquad_coefs(b0, b1, b2, a0, a1, a2, _ptr(c))
        return c

    @classmethod
    def parse_width(cls, s):
        t = C.c_int()
        ok = C.c_int()
        w = cls.lib().orc_parse_width(s.encode(), C.byref(t), C.byref(ok))
        return w, t.value, bool(ok.value)

    @classmethod
    def biquad_run(cls, c, x, m=None):
        """x: [frames, channels]; filters every channel in place with coefficients c; m: [channels,2] state."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        frames, ch = x.shape
        if m is None:
            m = np.zeros((ch, 2))
        c = np.ascontiguousarray(c, dtype=np.float64)
        for k in range(ch):
            cls.lib().orc_biquad_run(_ptr(c), m[k].ctypes.data, x.ctypes.data + 8 * k, frames, ch)
        return x, m

    @classmethod
    def per_channel(cls, kind, taps, x, *extra):
        """Run a single-channel convolver object (fir_direct|fir|fir_p|zita_equiv) over every column of x."""
        L = cls.lib()
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        frames, ch = x.shape
        taps = np.ascontiguousarray(taps, dtype=np.float64)
        if taps.ndim == 1:
            taps = np.repeat(taps[:, None], ch, axis=1)
        for k in range(ch):
            t = np.ascontiguousarray(taps[:, k])
            new = getattr(L, f"orc_{kind}_new")
            st = new(_ptr(t), len(t), *extra) if extra else new(_ptr(t), len(t))
            assert st
            getattr(L, f"orc_{kind}_run")(st, x.ctypes.data + 8 * k, frames, ch)
            getattr(L, f"orc_{kind}_free")(st)
        return x

    @classmethod
    def conv_full(cls, x, taps):
        x = np.ascontiguousarray(x, dtype=np.float64)
        taps = np.ascontiguousarray(taps, dtype=np.float64)
        y = np.zeros(len(x) + len(taps) - 1)
        cls.lib().orc_conv_full(_ptr(x), len(x), _ptr(taps), len(taps), _ptr(y))
        return y

    @classmethod
    def resample(cls, x, fs_in, fs_out, bw=0.939, block=2048):
        """Whole-stream resample incl. drain2; x: [frames, channels] -> [oframes, channels]."""
        L = cls.lib()
        x = np.ascontiguousarray(x, dtype=np.float64)
        frames, ch = x.shape
        outs = []
        for k in range(ch):
            st = L.orc_resample_new(fs_in, fs_out, bw)
            p = np.zeros(8, dtype=np.int32)
            L.orc_resample_params(st, _ptr(p))
            n, d = int(p[0]), int(p[1])
            col = np.ascontiguousarray(x[:, k])
            chunks = []
            pos = 0
            while pos < frames:
                nb = min(block, frames - pos)
                o = np.zeros(-(-nb * n // d) + 1)
                f = L.orc_resample_run(st, col.ctypes.data + 8 * pos, nb, 1, _ptr(o), 1)
                chunks.append(o[:f].copy())
                pos += nb
            scratch = np.zeros(block)
            while True:
                o = np.zeros(-(-block * n // d) + 1)
                f = L.orc_resample_drain(st, block, _ptr(scratch), _ptr(o))
                if f < 0:
                    break
                chunks.append(o[:f].copy())
            L.orc_resample_free(st)
            outs.append(np.concatenate(chunks) if chunks else np.zeros(0))
        return np.stack(outs, axis=1)

    @classmethod
    def hilbert_taps(cls, taps, angle_deg=-90.0):
        h = np.zeros(taps)
        cls.lib().orc_hilbert_taps(taps, angle_deg, _ptr(h))
        return h

    @classmethod
    def sgen_sine(cls, frames, channels, fs, freq, pos0=0):
        b = np.zeros((frames, channels))
        cls.lib().orc_sgen_sine(_ptr(b), frames, channels, fs, freq, pos0)
        return b


def _sgen_sweep(frames, channels, fs, f0, f1, total, pos0=0):
    b = np.zeros((frames, channels))
    Oracle.lib().orc_sgen_sweep(_ptr(b), frames, channels, fs, f0, f1, total, pos0)
    return b


def _sgen_delta(frames, channels, offset, pos0=0):
    b = np.zeros((frames, channels))
    Oracle.lib().orc_sgen_delta(_ptr(b), frames, channels, offset, pos0)
    return b


Oracle.sgen_sweep = staticmethod(_sgen_sweep)
Oracle.sgen_delta = staticmethod(_sgen_delta)


class RefChain:
    """The real reference's effects chain, in-process (oracle/ref_harness.c)."""

    _libs = {}

    @classmethod
    def so_path(cls, variant=""):
        return os.path.join(REF_DIR, f"libdspref{variant}.so")

    @classmethod
    def available(cls, variant=""):
        return os.path.exists(cls.so_path(variant))

    @classmethod
    def lib(cls, variant=""):
        if variant not in cls._libs:
            if variant == "_mkl":
                os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")
            L = C.CDLL(cls.so_path(variant))
            L.refh_chain_new.restype = C.c_void_p
            L.refh_chain_new.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
            L.refh_chain_free.argtypes = [C.c_void_p]
            L.refh_chain_reset.argtypes = [C.c_void_p]
            L.refh_chain_run.restype = _ss
            L.refh_chain_run.argtypes = [C.c_void_p, C.c_void_p, _ss, C.c_void_p]
            L.refh_chain_drain.restype = _ss
            L.refh_chain_drain.argtypes = [C.c_void_p, _ss, C.c_void_p]
            L.refh_chain_process.restype = _ss
            L.refh_chain_process.argtypes = [C.c_void_p, C.c_void_p, _ss, _ss, C.c_void_p, _ss]
            L.refh_chain_max_out_frames.restype = _ss
            L.refh_chain_max_out_frames.argtypes = [C.c_void_p, _ss]
            L.refh_chain_drain_frames.restype = _ss
            L.refh_chain_drain_frames.argtypes = [C.c_void_p]
            L.refh_chain_n_effects.restype = C.c_int
            L.refh_chain_n_effects.argtypes = [C.c_void_p]
            L.refh_chain_effect_name.restype = C.c_char_p
            L.refh_chain_effect_name.argtypes = [C.c_void_p, C.c_int]
            L.refh_bench.restype = C.c_double
            L.refh_bench.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, _ss, _ss, C.c_void_p]
            L.refh_ncpu.restype = C.c_int
            L.refh_set_loglevel.argtypes = [C.c_int]
            cls._libs[variant] = L
        return cls._libs[variant]

    def __init__(self, chain, fs, channels, directory=None, variant=""):
        self.L = self.lib(variant)
        ofs, och = C.c_int(), C.c_int()
        d = directory.encode() if directory else None
        self.h = self.L.refh_chain_new(chain.encode(), fs, channels, d, C.byref(ofs), C.byref(och))
        if not self.h:
            raise ValueError(f"reference rejected chain: {chain}")
        self.fs, self.channels = fs, channels
        self.ofs, self.ochannels = ofs.value, och.value

    def close(self):
        if self.h:
            self.L.refh_chain_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def effect_names(self):
        return [self.L.refh_chain_effect_name(self.h, i).decode() for i in range(self.L.refh_chain_n_effects(self.h))]

    def drain_frames(self):
        return self.L.refh_chain_drain_frames(self.h)

    def run(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        frames = x.shape[0]
        cap = max(self.L.refh_chain_max_out_frames(self.h, frames), frames)
        out = np.zeros((cap, self.ochannels))
        f = self.L.refh_chain_run(self.h, _ptr(x), frames, _ptr(out))
        assert f >= 0
        return out[:f].copy()

    def process(self, x, block=2048):
        """Whole stream incl. drain, like the CLI loop."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        frames = x.shape[0]
        cap = self.L.refh_chain_max_out_frames(self.h, frames) + self.L.refh_chain_max_out_frames(self.h, self.drain_frames() + 16 * block) + 64
        out = np.zeros((cap, self.ochannels))
        f = self.L.refh_chain_process(self.h, _ptr(x), frames, block, _ptr(out), cap)
        assert 0 <= f < cap
        return out[:f].copy()


def zita_contract(x, h):
    """The zita_convolver contract restated with an fp64 FFT (parity unpinned: libzita-convolver is absent): float32 input,
    float32 filter, exact convolution, float32 output (zita_convolver.cpp:44,53,110); the part_len frames of latency are
    removed by the host's end-of-chain alignment (:93-102), so the visible stream is the plain convolution.  x: [frames, ch],
    h: [taps].  Pinned against the oracle's direct-form zita_equiv at small sizes (tests/test_oracle_golden.py)."""
    from scipy.signal import fftconvolve
    xf = np.asarray(x, dtype=np.float64).astype(np.float32).astype(np.float64)
    hf = np.asarray(h, dtype=np.float64).astype(np.float32).astype(np.float64)
    y = np.stack([fftconvolve(xf[:, k], hf) for k in range(xf.shape[1])], axis=1)
    return y.astype(np.float32).astype(np.float64)


def rms(a):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt(np.mean(a * a))) if a.size else 0.0
