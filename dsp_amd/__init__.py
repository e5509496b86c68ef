"""dsp_amd -- MI355X-native run path for bmc0/dsp effects chains.

The product is the C-ABI shared library ``dsp_amd/libdsp_amd.so`` (HIP kernels for
gfx950 + the C++ host, built from ``dsp_amd/csrc``).  This package is the thin
Python host over it: ctypes bindings (``dsp_amd.lib``), the chain/batch wrappers
mirroring the reference's effects_chain API (``dsp_amd.chain``) and the
one-process-per-GPU stream sharding used by ``bench.py`` (``dsp_amd.shard``).
PyTorch is only plumbing here: device memory, streams, ``torch.distributed``.
"""
from .lib import load_library, library_path, LibraryMissing  # noqa: F401
from .chain import EffectsChain, BatchChain  # noqa: F401

__all__ = ["load_library", "library_path", "LibraryMissing", "EffectsChain", "BatchChain"]
