import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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


def load_golden(name):
    """Cases of tests/golden/<name>.json with the shared grids expanded."""
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
