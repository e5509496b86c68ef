"""ctypes binding of include/dsp_amd.h.  Fails loudly when the HIP extension is missing --
there is no CPU fallback and nothing here ever touches oracle/."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ssize_t = C.c_ssize_t


class LibraryMissing(RuntimeError):
    pass


def library_path():
    return os.path.join(_HERE, "libdsp_amd.so")


class _EffectInfo(C.Structure):
    _fields_ = [("name", C.c_char_p), ("usage", C.c_char_p), ("init", C.c_void_p), ("effect_number", C.c_int)]


class StreamInfo(C.Structure):
    _fields_ = [("fs", C.c_int), ("channels", C.c_int)]


class Effect(C.Structure):
    """struct effect, effect.h:39-59 (include/dsp_effect_abi.h)"""


_RUN = C.CFUNCTYPE(C.c_void_p, C.POINTER(Effect), C.POINTER(ssize_t), C.c_void_p, C.c_void_p)
Effect._fields_ = [
    ("prev", C.POINTER(Effect)), ("next", C.POINTER(Effect)), ("name", C.c_char_p),
    ("istream", StreamInfo), ("ostream", StreamInfo), ("channel_selector", C.c_void_p), ("flags", C.c_int),
    ("prepare", C.c_void_p), ("run", _RUN), ("reset", C.CFUNCTYPE(None, C.POINTER(Effect))), ("signal", C.c_void_p),
    ("plot", C.c_void_p), ("drain_samples", C.CFUNCTYPE(None, C.POINTER(Effect), C.POINTER(ssize_t))),
    ("drain2", _RUN), ("destroy", C.CFUNCTYPE(None, C.POINTER(Effect))),
    ("merge", C.CFUNCTYPE(C.c_int, C.POINTER(Effect), C.POINTER(Effect))), ("buffer_frames", C.c_void_p),
    ("channel_deps", C.c_void_p), ("channel_offsets", C.CFUNCTYPE(None, C.POINTER(Effect), C.POINTER(ssize_t), C.POINTER(ssize_t))),
    ("data", C.c_void_p),
]

PLUGIN_SYMBOLS = [
    "biquad_effect_init", "gain_effect_init", "remix_effect_init", "delay_effect_init", "fir_effect_init",
    "fir_p_effect_init", "resample_effect_init", "hilbert_effect_init", "zita_convolver_effect_init",
    "st2ms_effect_init", "crossfeed_effect_init",
    "fir_effect_init_with_filter", "fir_p_effect_init_with_filter", "zita_convolver_effect_init_with_filter",
    "delay_effect_init_int", "delay_effect_init_frac",
]
API_SYMBOLS = [
    "dspamd_version", "dspamd_last_error", "dspamd_device_count", "dspamd_set_device", "dspamd_set_loglevel",
    "dspamd_get_effect_info", "dspamd_plan_fir", "dspamd_chain_build", "dspamd_chain_run", "dspamd_chain_drain",
    "dspamd_chain_max_out_frames", "dspamd_chain_drain_frames", "dspamd_chain_reset", "dspamd_chain_destroy",
    "dspamd_chain_n_effects", "dspamd_chain_effect_name", "dspamd_batch_create", "dspamd_batch_out_fs",
    "dspamd_batch_out_channels", "dspamd_batch_max_out_frames", "dspamd_batch_drain_frames", "dspamd_batch_run", "dspamd_batch_run_strided",
    "dspamd_batch_drain", "dspamd_batch_reset", "dspamd_batch_destroy", "dspamd_batch_plan", "dspamd_batch_n_stages",
    "dspamd_sgen_sine", "dspamd_sgen_sweep", "dspamd_sgen_delta", "dspamd_digest", "dspamd_copy_probe", "dspamd_pcm_sample_bytes", "dspamd_pcm_read", "dspamd_pcm_write", "dspamd_profile_enable", "dspamd_profile_collect",
    "dspamd_batch_run_wire", "dspamd_batch_drain_wire", "dspamd_batch_wire_fused", "dspamd_plugin_counters",
]


def _preload_hip_runtime():
    """libdsp_amd.so carries no DT_NEEDED on a particular libamdhip64: a process must hold exactly ONE
    HIP runtime (a second one cannot open the GPU), and torch -- our plumbing for device memory,
    streams and RCCL -- bundles its own.  So: make torch's runtime global, fall back to /opt/rocm."""
    candidates = []
    try:
        import torch
        candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:  # torch absent: a plain C host would link -lamdhip64 itself
        pass
    candidates += [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "libamdhip64.so")]
    for c in candidates:
        if os.path.exists(c):
            C.CDLL(c, mode=C.RTLD_GLOBAL)
            return c
    raise LibraryMissing("no libamdhip64.so found (torch/lib or $ROCM_PATH/lib)")


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise LibraryMissing(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(hipcc --offload-arch=gfx950).  dsp_amd has no CPU fallback.")
    _preload_hip_runtime()
    L = C.CDLL(path)
    vp, cp, i = C.c_void_p, C.c_char_p, C.c_int
    sig = {
        "dspamd_version": (cp, []), "dspamd_last_error": (cp, []), "dspamd_device_count": (i, []),
        "dspamd_set_device": (i, [i]), "dspamd_set_loglevel": (None, [i]),
        "dspamd_get_effect_info": (C.POINTER(_EffectInfo), [cp]),
        "dspamd_plan_fir": (ssize_t, [cp, i, i, cp, i, i, vp, ssize_t, C.POINTER(ssize_t)]),
        "dspamd_chain_build": (vp, [cp, i, i, cp, C.POINTER(i), C.POINTER(i)]),
        "dspamd_chain_run": (ssize_t, [vp, vp, ssize_t, vp, ssize_t]),
        "dspamd_chain_drain": (ssize_t, [vp, ssize_t, vp, ssize_t]),
        "dspamd_chain_max_out_frames": (ssize_t, [vp, ssize_t]), "dspamd_chain_drain_frames": (ssize_t, [vp]),
        "dspamd_chain_reset": (None, [vp]), "dspamd_chain_destroy": (None, [vp]),
        "dspamd_chain_n_effects": (i, [vp]), "dspamd_chain_effect_name": (cp, [vp, i]),
        "dspamd_batch_create": (vp, [cp, i, i, i, ssize_t, cp]),
        "dspamd_batch_out_fs": (i, [vp]), "dspamd_batch_out_channels": (i, [vp]),
        "dspamd_batch_max_out_frames": (ssize_t, [vp, ssize_t]), "dspamd_batch_drain_frames": (ssize_t, [vp]),
        "dspamd_batch_run": (ssize_t, [vp, vp, ssize_t, vp, ssize_t, vp]),
        "dspamd_batch_run_strided": (ssize_t, [vp, vp, ssize_t, ssize_t, vp, ssize_t, vp]),
        "dspamd_batch_drain": (ssize_t, [vp, ssize_t, vp, ssize_t, vp]),
        "dspamd_batch_reset": (None, [vp, vp]), "dspamd_batch_destroy": (None, [vp]),
        "dspamd_batch_plan": (cp, [vp]), "dspamd_batch_n_stages": (i, [vp]),
        "dspamd_sgen_sine": (i, [vp, i, ssize_t, i, i, C.c_double, C.c_double, ssize_t, vp]),
        "dspamd_sgen_sweep": (i, [vp, i, ssize_t, i, i, C.c_double, C.c_double, C.c_double, ssize_t, ssize_t, vp]),
        "dspamd_sgen_delta": (i, [vp, i, ssize_t, i, ssize_t, ssize_t, ssize_t, vp]),
        "dspamd_digest": (i, [vp, i, ssize_t, ssize_t, i, vp, vp]),
        "dspamd_copy_probe": (i, [vp, vp, C.c_size_t, vp]),
        "dspamd_plugin_counters": (i, [C.POINTER(C.c_longlong), i]),
        "dspamd_pcm_sample_bytes": (C.c_size_t, [i]),
        "dspamd_pcm_read": (i, [i, vp, vp, ssize_t, vp]),
        "dspamd_pcm_write": (i, [i, vp, ssize_t, vp, i, ssize_t, i, i, ssize_t, vp, vp]),
        "dspamd_profile_enable": (None, [i]), "dspamd_profile_collect": (cp, []),
        "dspamd_batch_run_wire": (ssize_t, [vp, i, vp, ssize_t, ssize_t, i, vp, ssize_t, i, vp, vp]),
        "dspamd_batch_drain_wire": (ssize_t, [vp, ssize_t, i, vp, ssize_t, i, vp, vp]),
        "dspamd_batch_wire_fused": (i, [vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    init_sig = [C.POINTER(_EffectInfo), C.POINTER(StreamInfo), cp, cp, i, C.POINTER(cp)]
    for name in PLUGIN_SYMBOLS[:11]:
        f = getattr(L, name)
        f.restype = C.POINTER(Effect)
        f.argtypes = init_sig
    _LIB = L
    return L


PLUGIN_COUNTERS = ("wave_blocks", "mapped_blocks", "copied_blocks", "wave_launches", "wave_timeouts", "wave_off")


def plugin_counters():
    """how the plugin path has served its blocks in this process (dspamd_plugin_counters, include/dsp_amd.h)"""
    buf = (C.c_longlong * len(PLUGIN_COUNTERS))()
    n = load_library().dspamd_plugin_counters(buf, len(PLUGIN_COUNTERS))
    return {k: int(buf[j]) for j, k in enumerate(PLUGIN_COUNTERS[:n])}


def last_error():
    return load_library().dspamd_last_error().decode(errors="replace")
