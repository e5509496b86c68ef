"""Test configuration.

`pytest -m gpu` runs in ONE process again (round 6): the fault that made round 5 give every GPU test module a process of its own is found and gone
(host-buffer registrations, DESIGN.md section 5: 0 faults in 32 one-process runs of the suite without them, 3 in 13 with them), and
tests/test_gpu_soak.py builds and destroys 500 chains in the suite's own process.  The mechanism stays as an option: DSP_AMD_TESTS_ISOLATE_MARK=gpu hands
every module's selected tests to a child pytest and replays the child's reports, so a module starts on a fresh HIP context and a fault in one module
cannot take the others' results with it (a child that dies fails its test; CPU self-test: tests/test_isolation_cpu.py)."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

CHILD_REPORT = os.environ.get("DSP_AMD_TESTS_CHILD_REPORT")              # set in a child: where its reports go
ONE_PROCESS = os.environ.get("DSP_AMD_TESTS_ONE_PROCESS") == "1"
ISOLATE_MARK = os.environ.get("DSP_AMD_TESTS_ISOLATE_MARK")               # None: no isolation (the default); "gpu": the GPU modules; the CPU self-test isolates another mark


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "isolation_selftest: only for tests/test_isolation_cpu.py's inner run")


# ---- in a child: one JSON line per report
def pytest_runtest_logreport(report):
    if not CHILD_REPORT:
        return
    rec = {"nodeid": report.nodeid, "when": report.when, "outcome": report.outcome, "duration": getattr(report, "duration", 0.0),
           "longrepr": None, "sections": [(k, v[-20000:]) for k, v in report.sections]}
    if report.longrepr is not None:
        if report.outcome == "skipped" and isinstance(report.longrepr, tuple):
            rec["skip"] = [str(report.longrepr[0]), int(report.longrepr[1] or 0), str(report.longrepr[2])]
        rec["longrepr"] = str(report.longrepr)[-40000:]
    if hasattr(report, "wasxfail"):
        rec["wasxfail"] = report.wasxfail
    with open(CHILD_REPORT, "a") as f:
        f.write(json.dumps(rec) + "\n")


# ---- in the parent: a module's tests go to a child the first time one of them comes up
_modules = {}        # module path -> {"reports": {nodeid: [rec, ...]}, "rc": int, "tail": str, "died": {nodeid: (rc, tail)}}
_native_loaded = False


def _isolated(item):
    return bool(ISOLATE_MARK) and (not CHILD_REPORT) and (not ONE_PROCESS) and item.get_closest_marker(ISOLATE_MARK) is not None


class _Done:
    def __init__(self, returncode, stdout):
        self.returncode, self.stdout = returncode, stdout


def _run(cmd, cwd, env, limit=float(os.environ.get("DSP_AMD_TESTS_MODULE_SECONDS", "1500"))):
    """a child that outlives `limit` seconds (a hung kernel) is ended; it stays in this process's group, so whatever ends the suite ends it too"""
    proc = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, errors="replace")
    try:
        out, _ = proc.communicate(timeout=limit)
        return _Done(proc.returncode, out)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
        return _Done(-9, (out or "") + f"\n[tests/conftest.py: the module's test process was ended after {limit:.0f} s]\n")


def _run_module(item):
    global _native_loaded
    if ISOLATE_MARK == "gpu" and not _native_loaded:
        # (the library is in this process too: what `pytest -m gpu` loaded is what its children load)
        _native_loaded = True
        try:
            import dsp_amd
            dsp_amd.load_library()
        except Exception:  # the children report what is wrong with it, test by test
            pass
    module = item.nodeid.split("::")[0]
    ids = [i.nodeid for i in item.session.items if i.nodeid.split("::")[0] == module and _isolated(i)]
    maxfail = item.config.getoption("maxfail") or 0
    reports, rc, tail, todo, died = {}, 0, "", ids, {}
    for _ in range(6):
        fd, path = tempfile.mkstemp(prefix="dspamd_reports_", suffix=".jsonl")
        os.close(fd)
        env = dict(os.environ, DSP_AMD_TESTS_CHILD_REPORT=path)
        cmd = [sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "--rootdir", str(item.config.rootpath), "-m", ISOLATE_MARK, f"--maxfail={maxfail}"] + todo
        p = _run(cmd, str(item.config.rootpath), env)
        with open(path) as f:
            for line in f:
                rec = json.loads(line)
                reports.setdefault(rec["nodeid"], []).append(rec)
        os.unlink(path)
        rc, tail = p.returncode, p.stdout[-6000:]
        if rc in (0, 1, 2, 5):
            break                                   # pytest's own exit codes: the child got to its end (or to --maxfail)
        # the child died: the test it was in is the first one without a teardown report; the tests behind it get a process of their own
        left = [i for i in todo if not any(r["when"] == "teardown" for r in reports.get(i, []))]
        if not left:
            break
        died[left[0]] = (rc, tail)
        todo = left[1:]
        if not todo or maxfail:
            break
    _modules[module] = {"reports": reports, "rc": rc, "tail": tail, "died": died}
    return _modules[module]


@pytest.hookimpl(tryfirst=True)
def pytest_runtest_protocol(item, nextitem):
    if not _isolated(item):
        return None
    from _pytest.reports import TestReport
    module = item.nodeid.split("::")[0]
    res = _modules.get(module) or _run_module(item)
    recs = res["reports"].get(item.nodeid)
    ihook = item.ihook
    ihook.pytest_runtest_logstart(nodeid=item.nodeid, location=item.location)
    if item.nodeid in res["died"]:
        rc, tail = res["died"][item.nodeid]
        text = f"the module's test process died in this test (exit code {rc}); the end of its output:\n{tail}"
        recs = (recs or []) + [{"nodeid": item.nodeid, "when": "call" if recs else "setup", "outcome": "failed", "duration": 0.0, "longrepr": text, "sections": []}]
    elif not recs and not (item.config.getoption("maxfail") or 0):
        text = f"no report for this test from the module's test process (exit code {res['rc']}); the end of its output:\n{res['tail']}"
        recs = [{"nodeid": item.nodeid, "when": "setup", "outcome": "failed", "duration": 0.0, "longrepr": text, "sections": []}]
    for rec in recs or []:
        longrepr = rec.get("longrepr")
        if rec["outcome"] == "skipped" and rec.get("skip"):
            longrepr = tuple(rec["skip"])
        rep = TestReport(nodeid=item.nodeid, location=item.location, keywords={k: 1 for k in item.keywords}, outcome=rec["outcome"], longrepr=longrepr,
                         when=rec["when"], sections=[tuple(s) for s in rec.get("sections", [])], duration=rec.get("duration", 0.0))
        if "wasxfail" in rec:
            rep.wasxfail = rec["wasxfail"]
        ihook.pytest_runtest_logreport(report=rep)
    ihook.pytest_runtest_logfinish(nodeid=item.nodeid, location=item.location)
    return True
