"""The solver iteration itself against the reference's.  The reference's corrected rates are defined by where SciPy's
trust-region iteration stops (CorrectLambda.py:85,260,303,305); tests/golden/golden_traces.json.gz holds, for 21 golden
cases, every least_squares call the reference made with nfev, status and its trial points (make_golden.py wraps
scipy.optimize.least_squares while running the reference).  The HIP path reports the same per interval
(misti_last_solver_trace) and the two histories are compared solve by solve:

* determined cases: every solve has the reference's nfev and termination status, trial points agree to 1e-6;
* reference-indeterminate cases: the histories agree (nfev, status, trial points) up to a FIRST differing solve, which
  is reported (tools/trace_report.py -> profiles/); for --cpfit that solve is a long one (a runaway rate: the radius
  doubles for tens of iterations until |J^T f| < 1e-10 and rounding decides the last step)."""
import pytest

from conftest import load_golden
from parity import KNOWN_OUTSIDE, determined
from solver_trace_util import compare_case, load_traces

pytestmark = pytest.mark.gpu

TRACES = load_traces()
CASES = {c["name"]: c for f in ("golden_small", "golden_synthetic", "golden_sweep", "golden_campaign") for c in load_golden(f)}


@pytest.mark.parametrize("name", sorted(TRACES), ids=sorted(TRACES))
def test_iteration_history(name):
    case, ref = CASES[name], TRACES[name]
    r = compare_case(case, ref)
    fd = r["first_diff"]
    assert r["max_rel_before"] <= 1e-5, r                        # trial points of the solves before the first differing one
    if name in ("camp_s2_m35_c1", "camp_s2_m35_c9", "camp_s2_m35_c11", "camp_s2_m35_c21"):
        # round 2's documented outlier: the migrating interval of the default fit took the reference three evaluations and the
        # device two (its noise-free residual satisfied gtol one evaluation early).  Now the same iteration, solve by solve -
        # and the solver word says why (bit 24: went on past a gradient test only the reference's noisy residual fails)
        assert name not in KNOWN_OUTSIDE and fd is None and r["n_equal"] == r["n_solves"], r
        from solver_trace_util import hip_trace
        _, _, tr = hip_trace(case)
        sv = [s for s in ref["solves"] if s["site"] == "two_pop_ect"]
        assert len(sv) == 1 and (sv[0]["nfev"], sv[0]["status"]) == (3, 1)
        assert int(tr["noise"][0, sv[0]["t"]]) == 1 and int(tr["nfev"][0, sv[0]["t"]]) == 3
        return
    if determined(case["out"]):
        assert fd is None and r["n_equal"] == r["n_solves"], r   # the same iteration, solve by solve
        assert r["max_rel_before"] <= 1e-6
        return
    if fd is None:
        return
    # everything before the first differing solve agreed (compare_case stops counting max_rel there)
    solves = ref["solves"]
    k = next(i for i, s in enumerate(solves) if s["t"] == fd["t"])
    assert r["n_equal"] >= k, r
    if case["in"]["kw"].get("cpfit"):
        assert fd["ref"][0] >= 8 or fd["hip"][0] >= 8, r         # the flip sits inside a long (runaway) solve
