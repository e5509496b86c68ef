"""A minimal LADSPA host (TEST INFRASTRUCTURE): drives a LADSPA plugin library the way a host does -- dlopen,
ladspa_descriptor(i), instantiate, connect_port, run, cleanup -- through a ctypes mirror of the descriptor record restated in
oracle/ladspa_abi/ladspa.h.  Used on the two builds of the reference's UNMODIFIED ladspa_dsp.c (oracle/Makefile):
oracle/_ref/ladspa_dsp_ref.so (all reference effects, CPU) and oracle/_ref/ladspa_dsp_gpu.so (this repo's effects from
libdsp_amd.so).  ladspa_dsp reads its config files when the library is loaded (ladspa_dsp.c:384-401), so
LADSPA_DSP_CONFIG_PATH is set before dlopen, and each library gets a process of its own:

    python tests/ladspa_host.py LIB CONFIG_DIR LABEL FS BLOCKS IN.npy OUT.npy

IN.npy: float32 [frames][input ports]; BLOCKS: comma-separated run() sizes used in turn; OUT.npy: float32 [frames][output
ports].  Prints one JSON line: ports and the mean run() time per block size."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

PORT_INPUT, PORT_OUTPUT, PORT_CONTROL, PORT_AUDIO = 1, 2, 4, 8


class PortRangeHint(C.Structure):
    _fields_ = [("HintDescriptor", C.c_int), ("LowerBound", C.c_float), ("UpperBound", C.c_float)]


class Descriptor(C.Structure):
    pass


Handle = C.c_void_p
Descriptor._fields_ = [
    ("UniqueID", C.c_ulong), ("Label", C.c_char_p), ("Properties", C.c_int), ("Name", C.c_char_p), ("Maker", C.c_char_p),
    ("Copyright", C.c_char_p), ("PortCount", C.c_ulong), ("PortDescriptors", C.POINTER(C.c_int)),
    ("PortNames", C.POINTER(C.c_char_p)), ("PortRangeHints", C.POINTER(PortRangeHint)), ("ImplementationData", C.c_void_p),
    ("instantiate", C.CFUNCTYPE(Handle, C.POINTER(Descriptor), C.c_ulong)),
    ("connect_port", C.CFUNCTYPE(None, Handle, C.c_ulong, C.POINTER(C.c_float))),
    ("activate", C.CFUNCTYPE(None, Handle)),
    ("run", C.CFUNCTYPE(None, Handle, C.c_ulong)),
    ("run_adding", C.CFUNCTYPE(None, Handle, C.c_ulong)),
    ("set_run_adding_gain", C.CFUNCTYPE(None, Handle, C.c_float)),
    ("deactivate", C.CFUNCTYPE(None, Handle)),
    ("cleanup", C.CFUNCTYPE(None, Handle)),
]


def load(lib_path, config_dir):
    os.environ["LADSPA_DSP_CONFIG_PATH"] = config_dir
    os.environ.setdefault("LADSPA_DSP_LOGLEVEL", "NORMAL")
    lib = C.CDLL(lib_path)
    lib.ladspa_descriptor.restype = C.POINTER(Descriptor)
    lib.ladspa_descriptor.argtypes = [C.c_ulong]
    return lib


def descriptors(lib):
    out, i = [], 0
    while True:
        d = lib.ladspa_descriptor(i)
        if not d: return out
        out.append(d.contents)
        i += 1


def describe(d):
    kinds = [d.PortDescriptors[i] for i in range(d.PortCount)]
    return {"label": d.Label.decode(), "ports": int(d.PortCount),
            "inputs": sum(1 for k in kinds if k == (PORT_INPUT | PORT_AUDIO)),
            "outputs": sum(1 for k in kinds if k == (PORT_OUTPUT | PORT_AUDIO)),
            "names": [d.PortNames[i].decode() for i in range(d.PortCount)]}


def process(d, fs, blocks, x):
    info = describe(d)
    n_in, n_out = info["inputs"], info["outputs"]
    assert x.ndim == 2 and x.shape[1] == n_in and x.dtype == np.float32
    h = d.instantiate(C.pointer(d), fs)
    if not h: raise RuntimeError("instantiate() failed")
    if d.activate: d.activate(h)
    y = np.zeros((x.shape[0], n_out), dtype=np.float32)
    big = max(blocks)
    ports = [np.zeros(big, dtype=np.float32) for _ in range(n_in + n_out)]
    for i, p in enumerate(ports): d.connect_port(h, i, p.ctypes.data_as(C.POINTER(C.c_float)))
    times = {b: [] for b in blocks}
    pos, k = 0, 0
    while pos < x.shape[0]:
        b = blocks[k % len(blocks)]; k += 1
        n = min(b, x.shape[0] - pos)
        for c in range(n_in): ports[c][:n] = x[pos:pos + n, c]
        t0 = time.perf_counter()
        d.run(h, n)
        if n == b: times[b].append(time.perf_counter() - t0)
        for c in range(n_out): y[pos:pos + n, c] = ports[n_in + c][:n]
        pos += n
    if d.deactivate: d.deactivate(h)
    d.cleanup(h)
    # the first calls include allocation and kernel loading
    return y, {str(b): (float(np.mean(t[2:])) if len(t) > 2 else None) for b, t in times.items()}


def main(argv):
    lib_path, config_dir, label, fs, blocks, fin, fout = argv
    lib = load(lib_path, config_dir)
    ds = {describe(d)["label"]: d for d in descriptors(lib)}
    if label not in ds: raise SystemExit(f"no plugin labelled {label!r}: {sorted(ds)}")
    y, times = process(ds[label], int(fs), [int(b) for b in blocks.split(",")], np.load(fin))
    np.save(fout, y)
    print(json.dumps({"plugin": describe(ds[label]), "run_seconds": times}))


if __name__ == "__main__":
    main(sys.argv[1:])
