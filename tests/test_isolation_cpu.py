"""tests/conftest.py runs a GPU test module in a process of its own and replays its reports: the mechanism itself, on CPU, with another mark."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def inner(*args, **env):
    e = dict(os.environ, DSP_AMD_TESTS_ISOLATE_MARK="isolation_selftest")
    e.pop("DSP_AMD_TESTS_CHILD_REPORT", None)
    e.pop("DSP_AMD_TESTS_ONE_PROCESS", None)
    e.update(env)
    return subprocess.run([sys.executable, "-m", "pytest", "tests/isolation_sample.py", "-m", "isolation_selftest", "-p", "no:cacheprovider", *args],
                          cwd=ROOT, env=e, capture_output=True, text=True)


def test_reports_come_back_from_the_child():
    r = inner("-q", "-rs")
    assert r.returncode == 1, r.stdout[-2000:]
    assert re.search(r"1 failed, 5 passed, 1 skipped", r.stdout), r.stdout[-2000:]
    assert "the inner failure" in r.stdout and "the inner skip" in r.stdout


def test_x_stops_at_the_first_failure():
    r = inner("-q", "-x")
    assert r.returncode == 1 and re.search(r"1 failed, 3 passed", r.stdout), r.stdout[-2000:]


def test_a_child_that_dies_fails_its_tests_and_not_the_others():
    r = inner("-q", ISOLATION_SAMPLE_DIE="1")
    assert r.returncode == 1, r.stdout[-2000:]
    # the failure and the death; the test behind it ran in another process
    assert re.search(r"2 failed, 4 passed, 1 skipped", r.stdout), r.stdout[-2000:]
    assert "died in this test (exit code 7)" in r.stdout


def test_one_process_switch():
    r = inner("-q", DSP_AMD_TESTS_ONE_PROCESS="1")
    assert r.returncode == 1 and re.search(r"2 failed, 4 passed, 1 skipped", r.stdout), r.stdout[-2000:]   # (test_in_a_child is not in one)


def test_a_child_that_hangs_is_ended():
    r = inner("-q", ISOLATION_SAMPLE_HANG="1", DSP_AMD_TESTS_MODULE_SECONDS="3")
    assert r.returncode == 1, r.stdout[-2000:]
    assert re.search(r"2 failed, 4 passed, 1 skipped", r.stdout), r.stdout[-2000:]
    assert "was ended after 3 s" in r.stdout
