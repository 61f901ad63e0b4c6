"""Compare the HIP path's solver trace with the reference's (tests/golden/golden_traces.json.gz)."""
import contextlib
import gzip
import io
import json
import os

import numpy as np

from conftest import GOLDEN
from parity import engine_args

KIND_OF_SITE = {"two_pop_cp": 3, "two_pop_ect": 3, "no_migration": 2, "single_pop": 2}


def load_traces():
    out = {}
    for f in ("golden_traces.json.gz", "golden_campaign_traces.json.gz"):
        out.update({c["name"]: c for c in json.load(gzip.open(os.path.join(GOLDEN, f), "rt"))["cases"]})
    return out


def hip_trace(case):
    """Evaluate one golden case with the solver trace on: (llh, mirror object, trace dict)."""
    from misti_amd.engine import MigrationInference
    args, kw = engine_args(case["in"])
    with contextlib.redirect_stdout(io.StringIO()):
        m = MigrationInference(*args, **kw)
        m._engine.enable_solver_trace(True)
        llh = m.JAFSLikelihood(list(case["in"]["params"]))
        tr = m._engine.solver_trace(1, cand=0)
    return llh, m, tr


def compare_case(case, ref_trace, rel_iter=1e-6):
    """Walk the reference's solves in order.  Returns the number of solves whose (nfev, status) equal the device's,
    the first solve that differs (with the first trial point inside it that differs by more than rel_iter), and the
    largest relative difference between trial points of the solves before it."""
    llh, m, tr = hip_trace(case)
    split_in = case["in"]["split"]
    ins = int(split_in) if split_in % 1 else None          # the interval a fractional split shortens (not in the iterate record)
    n_equal, first, max_rel = 0, None, 0.0
    for sv in ref_trace["solves"]:
        t = sv["t"]
        hip = (int(tr["nfev"][0, t]), int(tr["status"][0, t]))
        kind = int(tr["kind"][0, t])
        ref = (sv["nfev"], sv["status"])
        same = hip == ref and kind == KIND_OF_SITE[sv["site"]]
        it_diff, rel_here, rel_at = None, 0.0, None
        if kind == 3 and sv["site"].startswith("two_pop") and t != ins:
            row = t if ins is None or t < ins else t - 1   # iterate rows are intervals of the shared grid
            dev = tr["iterates"][row]
            for i, x in enumerate(sv["trials"]):
                if i >= dev.shape[0] or not np.all(np.isfinite(dev[i])):
                    it_diff = i if it_diff is None else it_diff
                    break
                rel = float(np.max(np.abs(dev[i] - np.array(x)) / np.maximum(np.abs(np.array(x)), 1e-300)))
                if rel > rel_iter and it_diff is None:
                    it_diff, rel_at = i, rel
                if it_diff is None:
                    rel_here = max(rel_here, rel)
        if first is None:
            max_rel = max(max_rel, rel_here)
        if same and it_diff is None:
            n_equal += 1
        elif first is None:
            first = {"t": t, "site": sv["site"], "ref": ref, "hip": hip, "iter": it_diff, "rel_at_iter": rel_at}
    return {"llh": llh, "n_solves": len(ref_trace["solves"]), "n_equal": n_equal, "first_diff": first, "max_rel_before": max_rel}
