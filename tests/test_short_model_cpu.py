"""The one-trip convolver's transform on paper (scripts/conv_short_model.py): the index arithmetic of kernels_short.hip's three passes -- radix 32 / 16 / 16
at 8192 points, 32 / 32 / 16 at 16384, a thread's 32 points in the same positions in every pass -- reproduces numpy's FFT, and the exchange buffer's slots
pos + (pos >> 5) are conflict-free for every store and gather shape.  The kernel's own parity is tests/test_gpu_short.py; this is the part of its design that
can be checked without a GPU."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model():
    spec = importlib.util.spec_from_file_location("conv_short_model", os.path.join(ROOT, "scripts", "conv_short_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("log2n", [13, 14])
def test_exchange_slots_are_conflict_free(model, log2n):
    assert model.conflicts(log2n) == 1


@pytest.mark.parametrize("log2n,inverse", [(13, False), (13, True), (14, False), (14, True)])
def test_passes_reproduce_the_transform(model, log2n, inverse):
    assert model.transform(log2n, inverse) < 1e-13
