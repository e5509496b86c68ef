"""CPU-only checks of the product's host side: the C-ABI library loads, exports every symbol the
headers declare, and refuses to compute without a GPU (no silent CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import dsp_amd
    from dsp_amd.lib import API_SYMBOLS, PLUGIN_SYMBOLS
    L = dsp_amd.load_library()
    declared = set()
    for h in ("dsp_amd.h", "dsp_effect_abi.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        declared |= set(re.findall(r"\b((?:dspamd|biquad|gain|remix|delay|fir|fir_p|resample|hilbert|zita_convolver|st2ms|crossfeed)_[a-z0-9_]+)\s*\(", src))
    assert declared, "no declarations found"
    assert declared == set(API_SYMBOLS) | set(PLUGIN_SYMBOLS), declared ^ (set(API_SYMBOLS) | set(PLUGIN_SYMBOLS))
    for s in declared:
        assert hasattr(L, s), s


def test_registry_matches_reference_names():
    import dsp_amd
    L = dsp_amd.load_library()
    for name, num in [("lowpass", 7), ("eq", 13), ("linkwitz_transform", 17), ("biquad", 19), ("gain", 1), ("add", 3),
                      ("remix", 0), ("delay", 0), ("fir", 0), ("fir_p", 0), ("resample", 0), ("hilbert", 0), ("zita_convolver", 0),
                      ("st2ms", 1), ("ms2st", 2), ("crossfeed", 0)]:
        ei = L.dspamd_get_effect_info(name.encode())
        assert ei and ei.contents.name == name.encode() and ei.contents.effect_number == num
    assert not L.dspamd_get_effect_info(b"no_such_effect")


def test_no_cpu_fallback():
    import dsp_amd
    L = dsp_amd.load_library()
    if L.dspamd_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(ValueError, match="no HIP device"):
        dsp_amd.EffectsChain("gain -6", 48000, 2)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dsp_amd")):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in src and "oracle_api" not in src and "dsp_oracle" not in src and "libdspref" not in src, f


def test_filter_encodings_without_a_host_codec_layer_are_refused(tmp_path):
    """Raw PCM in an encoding this library does not decode itself (s24_3, u8 ...) is read through the reference host's
    fir_read_filter when the library sits inside that host (tests/test_gpu_endpoints.py); a stand-alone host has no codec
    layer and must refuse it with a message -- checked in a process of its own (no reference symbols in scope)."""
    import subprocess
    import sys
    f = tmp_path / "h.s24_3"
    f.write_bytes(bytes(range(90)))
    code = "\n".join([
        "import sys; sys.path.insert(0, %r)" % ROOT,
        "import ctypes, dsp_amd",
        "L = dsp_amd.load_library()",            # host-side planning needs no device (include/dsp_amd.h: dspamd_plan_fir)
        "d = ctypes.c_ssize_t()",
        "n = L.dspamd_plan_fir(b'fir -t pcm -e s24_3 -c 1 %s', 48000, 2, None, 0, 0, None, 0, ctypes.byref(d))" % str(f),
        "print('planned %d taps' % n if n >= 0 else 'refused: ' + L.dspamd_last_error().decode())",
    ])
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert "refused" in r.stdout and "codec layer" in r.stdout, r.stdout
