#!/usr/bin/env python3
"""How well is the reference's own llh determined by ITS arithmetic?  (needs /root/reference: run in the build container)

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/internal_noise.py [case names ...] > profiles/rNN_reference_internal_noise.txt

The input-perturbation study of make_golden.py moves times and rates by 2^-48 and re-runs the reference; every
intermediate then moves consistently with its inputs.  The default fit's residual (CorrectLambda.py:94-110) is
    T M^-1 exp(M T) p - M^-2 (exp(M T) - I) p,
whose two terms cancel to 1/|M T| of their size and whose second term carries the rounding error of scipy's expm
(absolute ~1e-16) multiplied by 1/|M T|^2.  Here that error is re-drawn directly: scipy.linalg.expm as the reference's
CorrectLambda module sees it returns its result with each entry moved by -1, 0 or +1 ulp at random, everything else
untouched, and the reference is re-run (`RUNS` seeds).  The largest relative change of llh is the case's
INTERNAL spread: what one ulp inside the reference's own matrix exponential does to its answer.

Third measurement (`--residual`, stored as `residual_spread`): one ulp of noise in the RESIDUAL VECTOR the reference hands to
scipy.optimize.least_squares - every value its residual functions (CorrectLambda.py:151-173, :237-251) return is moved by
-1, 0 or +1 ulp at random, nothing else.  A --cpfit solve whose rate has run away has a saturated residual: the gain ratio
actual/predicted reduction of its trust-region steps (scipy/optimize/_lsq/common.py:222-245) is then a quotient of two
rounding errors, and neither the input perturbations nor an ulp in expm re-draw the LAST rounding of that residual (the
subtraction of the target, CorrectLambda.py:141-144).  This one does."""
import contextlib
import io
import json
import os
import sys

import numpy

numpy.mat = numpy.asmatrix            # NumPy >= 2 removed the alias the reference imports
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import scipy.linalg                    # noqa: E402
import scipy.optimize                  # noqa: E402
import CorrectLambda as CL             # noqa: E402
import MigrationInference as MI        # noqa: E402
from conftest import load_golden       # noqa: E402

RUNS = 16


class NoisyLinalg:
    """scipy.linalg with expm moved by at most one ulp per entry."""
    def __init__(self, seed):
        self.rng = numpy.random.default_rng(seed)

    def __getattr__(self, name):
        return getattr(scipy.linalg, name)

    def expm(self, a):
        e = scipy.linalg.expm(a)
        k = self.rng.integers(-1, 2, e.shape)
        return numpy.where(k > 0, numpy.nextafter(e, numpy.inf), numpy.where(k < 0, numpy.nextafter(e, -numpy.inf), e))


class NoisyOptimize:
    """scipy.optimize whose least_squares sees every residual value moved by at most one ulp."""
    def __init__(self, seed):
        self.rng = numpy.random.default_rng(seed)

    def __getattr__(self, name):
        return getattr(scipy.optimize, name)

    def least_squares(self, fun, x0, *a, **kw):
        rng = self.rng

        def noisy(x, *fa, **fk):
            r = numpy.atleast_1d(numpy.asarray(fun(x, *fa, **fk), dtype=float))
            k = rng.integers(-1, 2, r.shape)
            return numpy.where(k > 0, numpy.nextafter(r, numpy.inf), numpy.where(k < 0, numpy.nextafter(r, -numpy.inf), r))
        return scipy.optimize.least_squares(noisy, x0, *a, **kw)


def run(i, noisy, noisy_opt=None):
    saved, saved_opt = CL.linalg, CL.optimize
    CL.linalg = noisy if noisy is not None else scipy.linalg
    CL.optimize = noisy_opt if noisy_opt is not None else scipy.optimize
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            m = MI.MigrationInference(list(i["times"]), [list(x) for x in i["lambdas"]], list(i["sfs"]), i["split"],
                                      [list(x) for x in i["mi"]], [list(x) for x in i["pu"]], **i["kw"])
            llh = m.JAFSLikelihood(list(i["params"]))
    finally:
        CL.linalg, CL.optimize = saved, saved_opt
    return float(llh) if numpy.isfinite(llh) else None


def study(c):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        base = run(c["in"], None)
        assert base == c["out"]["llh"], (c["name"], base, c["out"]["llh"])
        vals = [run(c["in"], NoisyLinalg(7000 + s)) for s in range(RUNS)]
    fin = [v for v in vals if v is not None]
    internal = max(abs(v - base) / abs(base) for v in fin) if fin else None
    return c["name"], {"internal_spread": internal, "runs": RUNS, "fails": RUNS - len(fin), "llh": vals}


def study_residual(c):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        base = run(c["in"], None)
        assert base == c["out"]["llh"], (c["name"], base, c["out"]["llh"])
        vals = [run(c["in"], None, NoisyOptimize(9000 + s)) for s in range(RUNS)]
    fin = [v for v in vals if v is not None]
    spread = max(abs(v - base) / abs(base) for v in fin) if fin else None
    return c["name"], {"residual_spread": spread, "residual_runs": RUNS, "residual_fails": RUNS - len(fin), "residual_llh": vals}


def main():
    import multiprocessing as mp
    from threadpoolctl import threadpool_limits
    names = [a for a in sys.argv[1:] if not a.startswith("-")]
    residual = "--residual" in sys.argv
    cases = [c for f in ("golden_small", "golden_synthetic", "golden_sweep", "golden_campaign") for c in load_golden(f) if c["out"]["llh"] is not None]
    if names:
        cases = [c for c in cases if c["name"] in names]
    path = os.path.join(HERE, "internal_noise.json")
    out = json.load(open(path))["cases"] if (names or residual) and os.path.exists(path) else {}
    print("%-32s %-8s %12s %12s %6s" % ("case", "fit", "input spread", "residual" if residual else "internal", "fails"))
    by_name = {c["name"]: c for c in cases}
    with threadpool_limits(1):
        with mp.get_context("fork").Pool(int(os.environ.get("PROCS", "8"))) as pool:
            for name, rec in pool.imap(study_residual if residual else study, cases):
                out.setdefault(name, {}).update(rec)       # the two studies share a record
                c = by_name[name]
                print("%-32s %-8s %12.3g %12.3g %6d" % (name, "cpfit" if c["in"]["kw"].get("cpfit") else "trueEPS" if c["in"]["kw"].get("trueEPS") else "default",
                                                        c["out"].get("spread") or 0, rec.get("residual_spread" if residual else "internal_spread") or 0,
                                                        rec.get("residual_fails" if residual else "fails")), flush=True)
    json.dump({"generator": "tests/golden/internal_noise.py", "runs": RUNS, "cases": out}, open(path, "w"))


if __name__ == "__main__":
    main()
