"""The solver iteration itself against the reference's.  The reference's corrected rates are defined by where SciPy's
trust-region iteration stops (CorrectLambda.py:85,260,303,305); tests/golden/golden_traces.json.gz holds, for 21 golden
cases, every least_squares call the reference made with nfev, status and its trial points (make_golden.py wraps
scipy.optimize.least_squares while running the reference).  The HIP path reports the same per interval
(misti_last_solver_trace) and the two histories are compared solve by solve:

* determined cases: every solve has the reference's nfev and termination status, trial points agree to 1e-6;
* reference-indeterminate cases: the histories agree (nfev, status, trial points) up to a FIRST differing solve, which
  is reported (tools/trace_report.py -> profiles/); for --cpfit that solve is a long one (a runaway rate: the radius
  doubles for tens of iterations until |J^T f| < 1e-10 and rounding decides the last step)."""
import numpy as np
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


# ---- round 4: the iteration itself at BASELINE size --------------------------------------------------------------------------
def _baseline_size_traces(name):
    import gzip
    import json
    import os
    from conftest import GOLDEN
    return {c["name"]: c for c in json.load(gzip.open(os.path.join(GOLDEN, name), "rt"))["cases"]}


def test_default_fit_iteration_at_baseline_size():
    """The default fit with migration at numT = 128 (configs 2 and 3; the 48 evenly spaced goldens of golden_default_fit.json that the
    reference completes): every least_squares call of the reference against the device's solver word of the same interval.  The
    reference's residual carries rounding noise there that decides single evaluations (its own llk moves by 1e-6 ...
    6e-3); `ect_noise_continues` imitates the noise's SIZE, so the histories cannot be equal solve for solve - what is held is how often
    (nfev, status) agree and that the device never differs by more than a few evaluations.  Measured on MI355X: profiles/r04_measured_guards.jsonl."""
    from parity import record
    traces = _baseline_size_traces("golden_default_fit_traces.json.gz")
    cases = [c for c in load_golden("golden_default_fit")[:48] if c["name"] in traces]
    assert len(cases) >= 40
    n_solves = n_equal = n_mig = n_mig_equal = 0
    worst = 0
    from solver_trace_util import KIND_OF_SITE, hip_trace
    for c in cases:
        llh, m, tr = hip_trace(c)
        if not np.isfinite(llh):
            continue
        for sv in traces[c["name"]]["solves"]:
            t = sv["t"]
            hip = (int(tr["nfev"][0, t]), int(tr["status"][0, t]))
            same = hip == (sv["nfev"], sv["status"]) and int(tr["kind"][0, t]) == KIND_OF_SITE[sv["site"]]
            n_solves += 1
            n_equal += same
            if sv["site"] == "two_pop_ect":
                n_mig += 1
                n_mig_equal += same
                worst = max(worst, abs(hip[0] - sv["nfev"]))
    record("default_fit_iteration_at_baseline_size", cases=len(cases), solves=n_solves, equal=n_equal, migrating=n_mig, migrating_equal=n_mig_equal, worst_nfev_difference=worst)
    assert n_mig > 2000
    from parity import pinned
    pinned(n_equal >= DEFAULT_FIT_EQUAL_MEASURED[0] * 0.97 * n_solves and n_mig_equal >= DEFAULT_FIT_EQUAL_MEASURED[1] * 0.97 * n_mig, ("(nfev, status) agreement", n_equal, n_solves, n_mig_equal, n_mig))
    assert n_equal >= 0.85 * n_solves and n_mig_equal >= 0.70 * n_mig, (n_equal, n_solves, n_mig_equal, n_mig)      # the floor the 256 fixed candidates are held to


def test_default_fit_iteration_256_fixed_candidates():
    """The same comparison on round 5's 256 candidates fixed in advance over configs 2, 3 and 5 (golden_default_fit_256.json): (nfev, status) of
    every least_squares call of the reference against the device's solver word of the same interval; recorded per workload."""
    from parity import record
    traces = _baseline_size_traces("golden_default_fit_256_traces.json.gz")
    cases = [c for c in load_golden("golden_default_fit_256") if c["name"] in traces]
    assert len(cases) >= 190
    from solver_trace_util import KIND_OF_SITE, hip_trace
    tot = {}
    for c in cases:
        llh, m, tr = hip_trace(c)
        if not np.isfinite(llh):
            continue
        wl = c["fullsize"]["workload"]
        t_ = tot.setdefault(wl, dict(cases=0, solves=0, equal=0, migrating=0, migrating_equal=0, worst_nfev_difference=0, stalls_predicted=0, stalls_confirmed=0))
        t_["cases"] += 1
        for sv in traces[c["name"]]["solves"]:
            t = sv["t"]
            hip = (int(tr["nfev"][0, t]), int(tr["status"][0, t]))
            same = hip == (sv["nfev"], sv["status"]) and int(tr["kind"][0, t]) == KIND_OF_SITE[sv["site"]]
            t_["solves"] += 1
            t_["equal"] += same
            if int(tr["stall"][0, t]):
                # the stall rule fired (trace bit 25): the device returned the starting point with status 3 after ONE evaluation and says so -
                # no made-up evaluation count any more; confirmed where the reference's own solve ended on xtol after >= 10 evaluations
                assert hip == (1, 3), hip
                t_["stalls_predicted"] += 1
                t_["stalls_confirmed"] += int(sv["status"] == 3 and sv["nfev"] >= 10)
            if sv["site"] == "two_pop_ect":
                t_["migrating"] += 1
                t_["migrating_equal"] += same
                if not int(tr["stall"][0, t]):
                    t_["worst_nfev_difference"] = max(t_["worst_nfev_difference"], abs(hip[0] - sv["nfev"]))
    record("default_fit_iteration_256", **tot)
    n_solves, n_equal = sum(v["solves"] for v in tot.values()), sum(v["equal"] for v in tot.values())
    n_mig, n_mig_equal = sum(v["migrating"] for v in tot.values()), sum(v["migrating_equal"] for v in tot.values())
    assert n_mig > 5000
    assert n_equal >= 0.85 * n_solves and n_mig_equal >= 0.70 * n_mig, (n_equal, n_solves, n_mig_equal, n_mig)
    stalls, confirmed = sum(v["stalls_predicted"] for v in tot.values()), sum(v["stalls_confirmed"] for v in tot.values())
    assert stalls == 0 or confirmed >= 0.5 * stalls, (stalls, confirmed)          # the rule's threshold is the 50 % point of the reference's own coin (profiles/r05_stall_calibration.txt)


# fraction of solves whose (nfev, status) equal the reference's: all solves / the migrating (two_pop_ect) ones; measured on MI355X, round 4
DEFAULT_FIT_EQUAL_MEASURED = (0.901, 0.790)    # 4 693 of 5 207 solves; 1 938 of 2 452 migrating solves; largest |nfev difference| 20


def test_cpfit_iteration_at_baseline_size():
    """The 43 full-size goldens of configs 3 and 5 (--cpfit; runaway rates, pulses, ancient sample): the histories agree solve for solve up
    to a first differing solve, which is a long (runaway) one - the flip the reference's own perturbed runs show as well."""
    traces = _baseline_size_traces("golden_fullsize_traces.json.gz")
    cases = [c for c in load_golden("golden_fullsize") if c["name"] in traces]
    assert len(cases) == 43
    n_all_equal = 0
    for c in cases:
        r = compare_case(c, traces[c["name"]])
        assert r["max_rel_before"] <= 1e-5, (c["name"], r)
        fd = r["first_diff"]
        if fd is None:
            n_all_equal += 1
            continue
        assert fd["ref"][0] >= 8 or fd["hip"][0] >= 8, (c["name"], r)          # the flip sits inside a long (runaway) solve
    from parity import record
    record("cpfit_iteration_at_baseline_size", cases=len(cases), identical_histories=n_all_equal)
