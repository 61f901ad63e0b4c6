#!/usr/bin/env python3
"""Table of HIP-vs-reference discrepancies over the golden vectors (run on the GPU box).

    python tools/parity_report.py [--md]

Columns: relative llk error, the tolerance of tests/parity.py, max relative JAFS
and lc errors, the reference's own sensitivity `sens` (see make_golden.py)."""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import load_golden                      # noqa: E402
from parity import determined, engine_args, llk_tol   # noqa: E402
from misti_amd.engine import MigrationInference       # noqa: E402


def main():
    rows = []
    for f in ("golden_small", "golden_synthetic"):
        for c in load_golden(f):
            o = c["out"]
            args, kw = engine_args(c["in"])
            with contextlib.redirect_stdout(io.StringIO()):
                m = MigrationInference(*args, **kw)
                llh = m.JAFSLikelihood(list(c["in"]["params"]))
            if o["llh"] is None:
                rows.append((c["name"], "-inf" if llh == -np.inf else "MISMATCH", "", "", "", "", m.status))
                continue
            if llh == -np.inf:
                rows.append((c["name"], "gpu -inf", "", "", "", "%.3g" % (o.get("sens") or -1), m.status))
                continue
            tol = llk_tol(o["llh"], c["in"]["sfs"], o["JAFS"], bool(kw.get("unfolded")))
            ej = np.max(np.abs(np.array(m.JAFS) / np.array(o["JAFS"]) - 1))
            el = np.max(np.abs(np.array(m.lc) / np.array(o["lc"]) - 1))
            rows.append((c["name"], "%.2e" % (abs(llh - o["llh"]) / abs(o["llh"])), "%.2e" % (tol / abs(o["llh"])),
                         "%.1e" % ej, "%.1e" % el, "%.3g" % (o.get("sens") or -1), "det" if determined(o) else "indet"))
    w = max(len(r[0]) for r in rows)
    print("%-*s %10s %10s %9s %9s %10s %s" % (w, "case", "llk rel", "llk tol", "JAFS rel", "lc rel", "sens", "class"))
    for r in rows:
        print("%-*s %10s %10s %9s %9s %10s %s" % ((w,) + r))


if __name__ == "__main__":
    main()
