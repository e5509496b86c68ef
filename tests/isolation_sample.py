"""Not collected by the suite (the name): the inner run of tests/test_isolation_cpu.py -- tests that pass, fail, skip and take their process down."""
import os

import pytest

pytestmark = pytest.mark.isolation_selftest


def test_in_a_child():
    assert os.environ.get("DSP_AMD_TESTS_CHILD_REPORT")


@pytest.mark.parametrize("chain", ["gain -3 lowpass 1k 0.707", "fir_p -t pcm {F}"])
def test_ids_with_spaces_and_braces(chain):
    assert chain


def test_fails():
    assert 1 + 1 == 3, "the inner failure"


def test_skips():
    pytest.skip("the inner skip")


def test_dies():
    if os.environ.get("ISOLATION_SAMPLE_DIE") == "1":
        os._exit(7)


def test_after_the_death():
    if os.environ.get("ISOLATION_SAMPLE_HANG") == "1":
        import time
        time.sleep(600)
