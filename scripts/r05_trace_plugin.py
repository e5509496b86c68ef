"""pytest plugin for the memory-fault hunt (scripts/r05_hunt_suite.sh): with DSP_AMD_TRACE_MEM=<file> set, every test's start goes into the library's
allocation trace, and the tests of test_gpu_short.py also record torch's device segments and the process's address map -- what a faulting address
printed by the runtime is looked up in.  Loaded with `-p r05_trace_plugin` (PYTHONPATH=scripts); not part of the suite."""
import os
import time


def pytest_runtest_setup(item):
    path = os.environ.get("DSP_AMD_TRACE_MEM")
    if not path:
        return
    with open(path, "a") as f:
        f.write(f"T {item.nodeid} {int(time.monotonic() * 1e6)}\n")
        if "test_gpu_short" in item.nodeid or "test_gpu_resident" in item.nodeid:
            try:
                import torch
                for seg in torch.cuda.memory_snapshot():
                    f.write(f"seg 0x{seg['address']:x} {seg['total_size']} {seg['segment_type']}\n")
            except Exception as e:  # pragma: no cover
                f.write(f"seg-error {e}\n")
            with open("/proc/self/maps") as m:
                maps = m.read()
            with open(path + ".maps", "a") as g:
                g.write(f"== {item.nodeid}\n{maps}\n")
