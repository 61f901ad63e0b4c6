#!/usr/bin/env python3
"""Solver traces of the HIP path against the reference's (run on the GPU box).

    python tools/trace_report.py > profiles/rNN_solver_traces.txt

For every golden case with a recorded reference trace (tests/golden/golden_traces.json.gz: every
scipy.optimize.least_squares call of the reference with nfev, status and trial points) the same case is evaluated
through the C ABI with the solver trace on, and the two iteration histories are compared interval by interval:
the first solve whose (nfev, status) differ, and inside it the first trial point that differs by more than 1e-6."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from solver_trace_util import compare_case, load_traces          # noqa: E402
from conftest import load_golden                                 # noqa: E402


def main():
    traces = load_traces()
    cases = {c["name"]: c for f in ("golden_small", "golden_synthetic", "golden_sweep", "golden_campaign") for c in load_golden(f)}
    print("%-32s %6s %6s %-22s %-26s %5s %9s %9s" % ("case", "solves", "equal", "first differing solve", "ref (nfev,status) | hip", "iter", "rel there", "rel before"))
    for name, tr in traces.items():
        r = compare_case(cases[name], tr)
        fd = r["first_diff"]
        print("%-32s %6d %6d %-22s %-26s %5s %9s %9.1e" % (
            name, r["n_solves"], r["n_equal"],
            "-" if fd is None else "t=%d (%s)" % (fd["t"], fd["site"]),
            "-" if fd is None else "(%d,%d) | (%d,%d)" % (fd["ref"] + fd["hip"]),
            "-" if fd is None or fd["iter"] is None else str(fd["iter"]),
            "-" if fd is None or fd.get("rel_at_iter") is None else "%.1e" % fd["rel_at_iter"],
            r["max_rel_before"]))
    print()
    print("solves / equal: least_squares calls of the reference / those whose nfev, status and every trial point (to 1e-6) the HIP path reproduces;")
    print("first differing solve: interval and call site; iter: first trial point differing by more than 1e-6 ('rel there'); rel before: largest")
    print("relative difference of any trial point in the solves before it.")


if __name__ == "__main__":
    main()
