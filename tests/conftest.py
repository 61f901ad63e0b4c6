import json
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# A crash of the host process (GPUTEST_r05: SIGSEGV, 0 passed) must name itself.  `-q` prints dots and the fault handler's dump ends
# with a 2.6 KB list of extension modules, so the tail a driver keeps of the log showed neither the test nor the native stack.  Now:
#  (1) the node id of the running test and the seconds since the session began go to the real stderr, unbuffered, BEFORE the
#      test's set-up starts (MISTI_TEST_TRACE=0 silences it), and to MISTI_TEST_TRACE_FILE (default gpurun_out/current_test.txt);
#  (2) tests/crashname.c (built here with gcc, test infrastructure) is installed as the handler faulthandler chains to: after the
#      Python stacks the log ENDS with the signal, the faulting thread's native backtrace as module(+offset) and the node id.
_TRACE_FILE = os.environ.get("MISTI_TEST_TRACE_FILE", os.path.join(ROOT, "gpurun_out", "current_test.txt"))
_T0 = time.time()
_crash = None


def _crash_reporter():
    """Build (if missing or stale) and load tests/_build/libcrashname.so; None when there is no compiler (the trace lines remain)."""
    import ctypes
    import shutil
    import subprocess
    src = os.path.join(ROOT, "tests", "crashname.c")
    out = os.path.join(ROOT, "tests", "_build", "libcrashname.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        cc = shutil.which("gcc") or shutil.which("cc")
        if cc is None:
            return None
        os.makedirs(os.path.dirname(out), exist_ok=True)
        tmp = "%s.%d.tmp" % (out, os.getpid())
        subprocess.run([cc, "-O1", "-g", "-fPIC", "-shared", "-o", tmp, src], check=True)
        os.replace(tmp, out)
    lib = ctypes.CDLL(out)
    lib.crashname_install.argtypes = [ctypes.c_int]
    lib.crashname_set.argtypes = [ctypes.c_char_p]
    return lib


@pytest.hookimpl(tryfirst=True)
def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    global _crash
    if os.environ.get("MISTI_TEST_TRACE", "1") == "0" or _crash is not None:
        return
    import faulthandler
    try:
        if faulthandler.is_enabled():          # PYTHONFAULTHANDLER / -X faulthandler: ours must be installed underneath it
            faulthandler.disable()             # (pytest's own plugin enables it again right after this hook)
        _crash = _crash_reporter()
        if _crash is not None and _crash.crashname_install(os.dup(2)) != 0:
            _crash = None
    except Exception as e:                     # the reporter is a convenience: never the reason a session does not start
        sys.stderr.write("crash reporter not installed: %r\n" % (e,))
        _crash = None


def pytest_runtest_logstart(nodeid, location):
    if os.environ.get("MISTI_TEST_TRACE", "1") == "0":
        return
    if _crash is not None:
        _crash.crashname_set(nodeid.encode())
    try:
        os.write(2, ("\n[test +%.1fs] %s\n" % (time.time() - _T0, nodeid)).encode())
    except OSError:
        pass
    try:
        os.makedirs(os.path.dirname(_TRACE_FILE), exist_ok=True)
        with open(_TRACE_FILE, "w") as f:
            f.write(nodeid + "\n")
    except OSError:
        pass


_NOISE = None


def _internal_noise():
    global _NOISE
    if _NOISE is None:
        p = os.path.join(GOLDEN, "internal_noise.json")
        _NOISE = json.load(open(p))["cases"] if os.path.exists(p) else {}
    return _NOISE


def _wide_spread():
    p = os.path.join(GOLDEN, "wide_spread.json")
    return json.load(open(p))["cases"] if os.path.exists(p) else {}


def load_golden(name, optional=False):
    """Cases of tests/golden/<name>.json with the shared grids expanded (`optional`: no such file -> no cases)."""
    if optional and not os.path.exists(os.path.join(GOLDEN, name + ".json")):
        return []
    d = json.load(open(os.path.join(GOLDEN, name + ".json")))
    grids = d.get("grids", {})
    noise = _internal_noise()
    wide = _wide_spread()
    for c in d["cases"]:
        w = wide.get(c["name"])
        if w is not None and c["out"].get("llh") is not None:        # tests/golden/wide_spread.py: clause 2b, the reference under 2^-44 perturbations
            c["out"]["spread_wide"] = w["spread_wide"]
        n = noise.get(c["name"])
        if n is not None and c["out"].get("llh") is not None:       # tests/golden/internal_noise.py: second measurement of the reference's indeterminacy
            c["out"]["internal_spread"], c["out"]["internal_fail"], c["out"]["internal_runs"] = n["internal_spread"], n["fails"], n["runs"]
        if "grid" in c["in"]:
            g = grids[c["in"]["grid"]]
            c["in"]["times"] = g["times"]
            c["in"]["lambdas"] = g["lambdas"]
    return d["cases"]


@pytest.fixture(scope="session")
def golden_small():
    return load_golden("golden_small")


@pytest.fixture(scope="session")
def golden_synthetic():
    return load_golden("golden_synthetic")
