"""CPU-only: the effect objects this library hands to a host behave like the reference's through the parts of the plugin ABI
that need no device -- init (argument parsing, NULL on error), flags, merge, prepare, channel_offsets, drain_samples,
channel_deps -- checked call for call against the real reference's own objects (oracle/_ref/libdspref.so, effect.h:39-59)."""
import ctypes as C

import numpy as np
import pytest

from oracle_api import RefChain

pytestmark = pytest.mark.skipif(not RefChain.available(), reason="oracle/_ref not present")


def _libs():
    import dsp_amd
    from dsp_amd.lib import Effect, StreamInfo, _EffectInfo, ssize_t
    A = dsp_amd.load_library()
    R = RefChain.lib()
    R.get_effect_info.restype = C.POINTER(_EffectInfo)
    R.get_effect_info.argtypes = [C.c_char_p]
    return A, R, Effect, StreamInfo, _EffectInfo, ssize_t


class Obj:
    """one effect of one library, driven through its vtable"""

    def __init__(self, lib, get_info, name, args, fs, channels, sel, Effect, StreamInfo, ssize_t):
        self.Effect, self.ssize_t, self.ch = Effect, ssize_t, channels
        ei = get_info(name.encode())
        assert ei, name
        init = C.CFUNCTYPE(C.POINTER(Effect), C.c_void_p, C.POINTER(StreamInfo), C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p))(ei.contents.init)
        si = StreamInfo(fs, channels)
        selb = bytes(1 if k in sel else 0 for k in range(channels))
        argv = (C.c_char_p * (len(args) + 1))(name.encode(), *[a.encode() for a in args])
        self.e = init(C.cast(ei, C.c_void_p), C.byref(si), selb, None, len(args) + 1, argv)
        self.libc = C.CDLL(None)

    def ok(self):
        return bool(self.e)

    def call_prepare(self):
        p = self.e.contents.prepare
        return C.CFUNCTYPE(C.c_int, C.POINTER(self.Effect))(p)(self.e) if p else 0

    def merge(self, other):
        m = self.e.contents.merge
        return m(self.e, other.e) if m else 0

    def offsets(self):
        lat = (self.ssize_t * self.ch)(); req = (self.ssize_t * self.ch)()
        f = self.e.contents.channel_offsets
        if f:
            f(self.e, lat, req)
        return list(lat), list(req)

    def drain(self):
        d = (self.ssize_t * self.ch)()
        f = self.e.contents.drain_samples
        if f:
            f(self.e, d)
        return list(d)

    def deps(self):
        f = self.e.contents.channel_deps
        if not f:
            return None
        rows = [(C.c_char * self.ch)() for _ in range(self.ch)]
        for k in range(self.ch):
            rows[k][k] = 1                                    # the host pre-sets identity (effects_chain.c:687-700)
        ptrs = (C.c_void_p * self.ch)(*[C.addressof(r) for r in rows])
        C.CFUNCTYPE(None, C.POINTER(self.Effect), C.c_void_p)(f)(self.e, ptrs)
        return [[int(r[j] != b"\0") for j in range(self.ch)] for r in rows]

    def has_run(self):
        return bool(C.cast(self.e.contents.run, C.c_void_p).value)

    def free(self):
        if self.e:
            if C.cast(self.e.contents.destroy, C.c_void_p).value:     # destroy_effect(), effect.c:78-85: all callbacks may be NULL
                self.e.contents.destroy(self.e)
            self.libc.free(self.e)
            self.e = None


def pair(name, args, fs=48000, channels=4, sel=None):
    A, R, Effect, StreamInfo, _EffectInfo, ssize_t = _libs()
    sel = set(range(channels)) if sel is None else set(sel)
    a = Obj(A, A.dspamd_get_effect_info, name, args, fs, channels, sel, Effect, StreamInfo, ssize_t)
    r = Obj(R, R.get_effect_info, name, args, fs, channels, sel, Effect, StreamInfo, ssize_t)
    return a, r


@pytest.mark.parametrize("args,sel", [
    (["10S"], None), (["-3S"], [1]), (["1.5m"], [0, 2]),
    (["-f", "0.3S"], None), (["-f1", "2.7S"], [0]), (["-f2", "7.25S"], [1, 3]), (["-f", "1.25m"], [0]), (["-f", "-2.5S"], [2]),
    (["-f3", "0.4S"], None), (["-f5", "2.3S"], [0, 1]), (["-f8", "0.77S"], [3]), (["-f10", "12.5S"], None),
])
def test_delay_offsets_and_drain(args, sel):
    a, r = pair("delay", args, sel=sel)
    assert a.ok() and r.ok()
    assert a.e.contents.flags == r.e.contents.flags
    assert a.call_prepare() == r.call_prepare() == 0
    assert a.offsets() == r.offsets()
    assert a.drain() == r.drain()
    a.free(); r.free()


def test_delay_merge_adds_amounts_and_takes_the_highest_order():
    objs = []
    for args in (["3S"], ["-f1", "0.4S"], ["-f", "0.35S"]):
        objs.append(pair("delay", args, channels=2))
    (a0, r0), rest = objs[0], objs[1:]
    for a, r in rest:
        assert a0.merge(a) == r0.merge(r) == 1
    assert a0.call_prepare() == r0.call_prepare() == 0
    # 3 + 0.4 + 0.35 = 3.75 samples; orders max(0, 1, 0) = 1 (an explicit order beats the default of the others):
    # integer part 3, first-order all-pass for 0.75, one more sample to drain
    assert a0.offsets() == r0.offsets() == ([0, 0], [3, 3])
    assert a0.drain() == r0.drain() == [1, 1]
    for a, r in objs:
        a.free(); r.free()


@pytest.mark.parametrize("name,args,sel,good", [
    ("st2ms", [], [0, 1], True), ("ms2st", [], [1, 3], True), ("st2ms", [], [0, 1, 2], False), ("st2ms", ["x"], [0, 1], False),
    ("crossfeed", ["700", "4.5"], [0, 2], True), ("crossfeed", ["700"], [0, 2], False), ("crossfeed", ["700", "-1"], [0, 2], False),
    ("crossfeed", ["30k", "3"], [0, 1], False),
    ("delay", ["-f99", "1S"], None, False), ("delay", ["-f", "abc"], None, False), ("delay", [], None, False),
])
def test_pair_effects_init_and_deps(name, args, sel, good):
    a, r = pair(name, args, sel=sel)
    assert a.ok() == r.ok() == good
    if good:
        assert a.e.contents.flags == r.e.contents.flags
        assert a.deps() == r.deps()
        assert a.has_run() and r.has_run()
    a.free(); r.free()


def test_zero_delays_are_dropped_by_the_host():
    # an effect that does nothing comes back with run == NULL and the host drops it (effects_chain.c:586-590)
    for args in (["0S"], ["-f", "0S"]):
        a, r = pair("delay", args)
        assert a.ok() and r.ok() and not a.has_run() and not r.has_run()
        a.free(); r.free()


COEFS40 = "coefs:" + ",".join(f"{0.9 ** i * (1 if i % 3 else -1):.6f}" for i in range(40))


@pytest.mark.parametrize("name,args,sel,channels", [
    ("lowpass", ["1k", "0.707"], None, 2), ("eq", ["2k", "1.5", "4"], [1], 3), ("highshelf", ["8k", "0.7s", "-3"], None, 2),
    ("lowpass", ["-r", "1k", "0.707"], [0], 2), ("highpass", ["-r60", "40", "0.5"], None, 2),
    ("gain", ["-6"], [0, 2], 4), ("mult", ["0.5"], None, 2), ("add", ["0.01"], [1], 2),
    ("remix", ["0,1", "2", ".", "1,2,3"], None, 4), ("remix", ["1", "0"], [0, 1], 3),
    ("fir", [COEFS40], None, 2), ("fir", ["coefs:1,0.5,0.25"], [0], 2), ("fir", ["-a", COEFS40], None, 2),
    ("fir_p", [COEFS40], [1], 3), ("fir_p", ["coefs:1,2,3/4,5,6"], [0, 1], 2),
    ("hilbert", ["127"], None, 2), ("hilbert", ["-p", "-c", "255"], [0], 2), ("hilbert", ["-a", "45", "63"], None, 1),
    ("resample", ["96k"], None, 2), ("resample", ["0.9", "44.1k"], None, 2), ("resample", ["48k"], None, 2),
])
def test_effect_objects_match_the_reference(name, args, sel, channels):
    a, r = pair(name, args, channels=channels, sel=sel)
    assert a.ok() and r.ok(), (name, args)
    assert a.has_run() == r.has_run()
    if a.has_run():
        ea, er = a.e.contents, r.e.contents
        assert (ea.ostream.fs, ea.ostream.channels, ea.istream.fs, ea.istream.channels) == (er.ostream.fs, er.ostream.channels, er.istream.fs, er.istream.channels)
        assert ea.flags == er.flags, (ea.flags, er.flags)
        a.ch = r.ch = max(ea.istream.channels, ea.ostream.channels)
        assert a.call_prepare() == r.call_prepare() == 0
        assert a.offsets() == r.offsets()
        assert a.drain() == r.drain()
        assert bool(C.cast(ea.drain2, C.c_void_p).value) == bool(C.cast(er.drain2, C.c_void_p).value)
        # which callbacks exist (effect.h:39-59): a host branches on every one of these being NULL or not
        # (plot_effects_chain refuses a chain with a plot-less effect, effects_chain.c:1130-1133)
        for cb in ("merge", "plot", "reset", "channel_deps", "channel_offsets", "drain_samples", "prepare", "signal", "buffer_frames"):
            assert bool(C.cast(getattr(ea, cb), C.c_void_p).value) == bool(C.cast(getattr(er, cb), C.c_void_p).value), (name, cb)
    a.free(); r.free()


@pytest.mark.parametrize("name,args", [
    ("lowpass", ["1k"]), ("lowpass", ["30k", "0.7"]), ("eq", ["1k", "1q", "x"]), ("gain", []), ("gain", ["abc"]),
    ("remix", []), ("fir", []), ("fir", ["coefs:"]), ("fir_p", ["coefs:1,2/3"]), ("hilbert", ["128"]), ("hilbert", ["-a"]),
    ("resample", []), ("resample", ["1.5", "96k"]), ("resample", ["-5k"]),
])
def test_odd_arguments_same_verdict(name, args):
    # whatever the reference makes of these (most are errors -> NULL, a few are accepted), the library does the same
    a, r = pair(name, args, channels=2)
    assert a.ok() == r.ok(), (name, args, a.ok(), r.ok())
    a.free(); r.free()


def test_wav_filter_file_init_cpu(tmp_path):
    # filter ingestion is host-side: a RIFF/WAVE file gives a `fir` effect with the file's length in its drain accounting
    # (fir.c:180-187), a damaged header or an unsupported sample format gives NULL -- no device needed
    import struct
    A, R, Effect, StreamInfo, _EffectInfo, ssize_t = _libs()
    A.dspamd_get_effect_info.restype = C.POINTER(_EffectInfo)
    A.dspamd_get_effect_info.argtypes = [C.c_char_p]
    q = (np.arange(1, 101) * 100).astype("<i2")
    body = struct.pack("<HHIIHH", 1, 1, 48000, 96000, 2, 16)
    wav = b"RIFF" + struct.pack("<I", 4 + 8 + len(body) + 8 + q.nbytes) + b"WAVE" + b"fmt " + struct.pack("<I", len(body)) + body + b"data" + struct.pack("<I", q.nbytes) + q.tobytes()
    good = tmp_path / "h.wav"; good.write_bytes(wav)
    lit = "coefs:" + ",".join("%.17g" % (v / 32768.0) for v in q)
    a = Obj(A, A.dspamd_get_effect_info, "fir", [str(good)], 48000, 2, {0, 1}, Effect, StreamInfo, ssize_t)
    b = Obj(A, A.dspamd_get_effect_info, "fir", [lit], 48000, 2, {0, 1}, Effect, StreamInfo, ssize_t)
    r = Obj(R, R.get_effect_info, "fir", [lit], 48000, 2, {0, 1}, Effect, StreamInfo, ssize_t)
    assert a.ok() and b.ok() and r.ok()
    assert a.drain() == b.drain() == r.drain()
    assert a.offsets() == b.offsets() == r.offsets()
    adpcm = wav[:20] + struct.pack("<H", 2) + wav[22:]            # format tag 2: not read
    bad = tmp_path / "adpcm.wav"; bad.write_bytes(adpcm)
    assert not Obj(A, A.dspamd_get_effect_info, "fir", [str(bad)], 48000, 2, {0, 1}, Effect, StreamInfo, ssize_t).ok()
    trunc = tmp_path / "trunc.wav"; trunc.write_bytes(wav[:30])
    assert not Obj(A, A.dspamd_get_effect_info, "fir", [str(trunc)], 48000, 2, {0, 1}, Effect, StreamInfo, ssize_t).ok()
    assert not Obj(A, A.dspamd_get_effect_info, "fir", ["-r", "48k", str(good).replace("h.wav", "missing.wav")], 48000, 2, {0, 1}, Effect, StreamInfo, ssize_t).ok()
    # a container whose rate differs from the stream's: refused (fir_util.c:103-109, config->p.fs = istream->fs by default, :130)
    # unless `-r any` was given (:148-150) -- for every consumer of fir_read_filter
    body441 = struct.pack("<HHIIHH", 1, 1, 44100, 88200, 2, 16)
    wav441 = b"RIFF" + struct.pack("<I", 4 + 8 + len(body441) + 8 + q.nbytes) + b"WAVE" + b"fmt " + struct.pack("<I", len(body441)) + body441 + b"data" + struct.pack("<I", q.nbytes) + q.tobytes()
    f441 = tmp_path / "h441.wav"; f441.write_bytes(wav441)
    for eff in ("fir", "fir_p", "zita_convolver"):
        assert not Obj(A, A.dspamd_get_effect_info, eff, [str(f441)], 48000, 2, {0, 1}, Effect, StreamInfo, ssize_t).ok(), eff
        assert not Obj(A, A.dspamd_get_effect_info, eff, ["-r", "48k", str(f441)], 48000, 2, {0, 1}, Effect, StreamInfo, ssize_t).ok(), eff
        assert Obj(A, A.dspamd_get_effect_info, eff, ["-r", "any", str(f441)], 48000, 2, {0, 1}, Effect, StreamInfo, ssize_t).ok(), eff
        assert Obj(A, A.dspamd_get_effect_info, eff, [str(f441)], 44100, 2, {0, 1}, Effect, StreamInfo, ssize_t).ok(), eff


def test_selector_grammar_differential():
    # the selector parser is this library's own (csrc/host_util.cpp); the grammar and the verdicts are the reference's
    # (util.c parse_selector / parse_selector_masked): random strings through `remix`, whose arguments are selectors counted
    # over the channels the block selector lets through -- same accept / reject, same dependency matrix
    rng = np.random.Generator(np.random.PCG64(4242))
    alphabet = list("0123456789") + [",", "-", "-", ",", "x", " ", "10", "11"]
    handmade = ["", "-", "0", "0-", "-2", "1-2", "2-1", "0,", ",0", "0,,1", "--", "1--2", "0-1-2", "9", "0-9", "3,2,1", "1,1", "-0", "01", "007-", "0-,1", "-,-"]      # (numbers beyond int: the reference wraps them through atoi, this library refuses them)
    def item():
        a, b = int(rng.integers(0, 4)), int(rng.integers(0, 5))
        return str(rng.choice([f"{a}", f"{a}-{b}", f"{a}-", f"-{b}", "-"]))
    strings = handmade + ["".join(rng.choice(alphabet, size=int(rng.integers(1, 7)))) for _ in range(300)]
    strings += [",".join(item() for _ in range(int(rng.integers(1, 4)))) for _ in range(300)]      # mostly well-formed
    accepted = 0
    for i, s in enumerate(strings):
        channels = int(rng.integers(1, 7))
        sel = sorted(set(int(c) for c in rng.integers(0, channels, size=int(rng.integers(1, channels + 1)))))
        args = [s] if s else ["."]
        if rng.integers(0, 2):
            args.append(str(rng.integers(0, len(sel))))
        a, r = pair("remix", args, channels=channels, sel=sel)
        assert a.ok() == r.ok(), (s, channels, sel, a.ok(), r.ok())
        if a.ok():
            accepted += 1
            assert (a.e.contents.ostream.channels, a.e.contents.flags) == (r.e.contents.ostream.channels, r.e.contents.flags), s
            a.ch = r.ch = max(channels, a.e.contents.ostream.channels)
            assert a.deps() == r.deps(), (s, channels, sel)
        a.free(); r.free()
    assert accepted > 50


def test_option_scanner_differential():
    # the option scanner is this library's own; the dialect is dsp_getopt's (util.c): clustered flags, attached / detached /
    # optional arguments, "--", unknown letters, a missing argument -- random option words in front of a valid filter
    rng = np.random.Generator(np.random.PCG64(77))
    words = ["-a", "-a3S", "-a-2S", "-B", "-L", "-N", "-BL", "-LNa", "-c", "1", "-c1", "-t", "pcm", "-tpcm", "-e", "double", "-edouble",
             "-r", "48k", "-r48k", "-rany", "--", "-x", "-", "-Bx", "-c", "-e", "-:", "-a:", "-Be", "double"]
    same_init = 0
    for _ in range(300):
        opts = [str(w) for w in rng.choice(words, size=int(rng.integers(0, 5)))]
        for name in ("fir", "fir_p"):
            a, r = pair(name, opts + [COEFS40], channels=2)
            assert a.ok() == r.ok(), (name, opts, a.ok(), r.ok())
            if a.ok():
                same_init += 1
                assert a.offsets() == r.offsets(), (name, opts)
                assert a.drain() == r.drain(), (name, opts)
            a.free(); r.free()
    for opts in (["-f"], ["-f3"], ["-f", "3"], ["-f0"], ["-fx"], ["-f2", "--"], ["--", "-f"], ["-q"]):
        a, r = pair("delay", opts + ["2.5S"], channels=2)
        assert a.ok() == r.ok() or opts == ["-q"], opts      # (-m/-M/-b/-q: modulated delay, refused by this backend on purpose)
        a.free(); r.free()
    assert same_init > 40
