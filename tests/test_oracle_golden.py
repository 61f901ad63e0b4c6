"""The CPU oracle against the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle."""
import math
import warnings

import numpy as np
import pytest

from conftest import load_golden
from oracle.misti_oracle import OracleModel, TWO_POP, ONE_POP

SMALL = load_golden("golden_small")
SYNTH = load_golden("golden_synthetic")
SWEEP = load_golden("golden_sweep")
CAMPAIGN = load_golden("golden_campaign")
# numT = 128 cases run through the reference in round 4 (tests/golden/make_fullsize.py); ~2 s of oracle each: every fourth / eighth one here
FULLSIZE = load_golden("golden_fullsize")[::4]
DEFAULT_FIT = load_golden("golden_default_fit")[::8]
FULLSIZE_R05 = load_golden("golden_fullsize_r05")[::4] + load_golden("golden_config5_default_sample")[::4] + load_golden("golden_config2b")[::4] + load_golden("golden_config2c")[::6] + load_golden("golden_config3b")[::3] + load_golden("golden_config2n255")[::10] + load_golden("golden_config2u")[::6] + load_golden("golden_config2m")[::4] + load_golden("golden_config2f") + load_golden("golden_config2b_allchains")[5::16] + load_golden("golden_config3b_fixed64")[3::16] + load_golden("golden_config5b_fixed64")[5::16] + load_golden("golden_config5b_default_fixed64", optional=True)[9::32] + load_golden("golden_config3b_default_fixed64", optional=True)[13::32]
DEFAULT_FIT_256 = load_golden("golden_default_fit_256")[7::16]        # round 5's 256 fixed candidates (configs 2, 3 and 5, default fit): 16 of them here
# the oracle restates the reference operation by operation on the same SciPy, so
# agreement is at rounding level; 1e-12 leaves room for a different BLAS build
RTOL = 1e-12


def run(case):
    i = case["in"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = OracleModel(i["times"], i["lambdas"], i["sfs"], i["split"], i["mi"], i["pu"], **i["kw"])
        llh = m.jafs_likelihood(i["params"])
    return m, llh


def check(case):
    m, llh = run(case)
    o = case["out"]
    assert m.numT == o["numT"] and m.splitT == o["splitT"]
    assert m.llh_const == pytest.approx(o["llh_const"], rel=1e-15)
    if o["llh"] is None:
        assert llh == -np.inf
        want = {"Hit negative value of migration rate": 1, "Lambda correction failed": 2}[o["stdout"][0]]
        assert m.status == want
        return
    assert llh == pytest.approx(o["llh"], rel=RTOL)
    np.testing.assert_allclose(m.JAFS, o["JAFS"], rtol=RTOL)
    np.testing.assert_allclose(np.array(m.lc, dtype=float), np.array(o["lc"]), rtol=RTOL)
    if "Pr" in o:
        np.testing.assert_allclose(np.array(m.Pr, dtype=float), np.array(o["Pr"]), rtol=RTOL, atol=1e-300)


@pytest.mark.parametrize("case", SMALL, ids=[c["name"] for c in SMALL])
def test_small(case):
    check(case)


@pytest.mark.parametrize("case", SYNTH, ids=[c["name"] for c in SYNTH])
def test_synthetic(case):
    check(case)


@pytest.mark.parametrize("case", SWEEP, ids=[c["name"] for c in SWEEP])
def test_sweep(case):
    """The README's four-band st x mc sweep (README.md:110-115), one reference run per grid point."""
    check(case)


@pytest.mark.parametrize("case", CAMPAIGN, ids=[c["name"] for c in CAMPAIGN])
def test_campaign_worst(case):
    """The random campaign's worst candidates (tools/random_campaign.py), run through the reference: the oracle the
    campaign is judged against reproduces the reference on them."""
    check(case)


@pytest.mark.parametrize("case", FULLSIZE_R05, ids=[c["name"] for c in FULLSIZE_R05])
def test_fullsize_outliers_round5(case):
    """Round 5's reference-run studies of full-grid outliers: the oracle reproduces the reference on them."""
    check(case)


@pytest.mark.parametrize("case", DEFAULT_FIT_256, ids=[c["name"] for c in DEFAULT_FIT_256])
def test_default_fit_256(case):
    """Round 5's default-fit candidates fixed in advance over configs 2, 3 and 5 (numT = 128, ancient sample and pulse included)."""
    check(case)


@pytest.mark.parametrize("case", FULLSIZE, ids=[c["name"] for c in FULLSIZE])
def test_fullsize_outliers(case):
    """Full-size candidates (configs 3 and 5) run through the reference: the oracle reproduces the reference there too."""
    check(case)


@pytest.mark.parametrize("case", DEFAULT_FIT, ids=[c["name"] for c in DEFAULT_FIT])
def test_default_fit_at_baseline_size(case):
    """The default fit with migration at numT = 128 through the reference: the oracle reproduces it (values AND failures)."""
    check(case)


def test_perturbation_records_are_consistent():
    """`sens` (round 1's three perturbations) and `spread` (all of them) describe the same runs."""
    from parity import PERTURB, N_KINDS_BASE, N_KINDS_DEEP, SENS_DETERMINED
    n_deep = 0
    for c in SMALL + SYNTH + SWEEP:
        o = c["out"]
        if o["llh"] is None:
            assert 0 <= o["pert_finite"] <= N_KINDS_BASE
            continue
        vals = o["pert_llh"]
        deep = o["sens"] is None or o["sens"] >= SENS_DETERMINED
        assert len(vals) == (N_KINDS_DEEP if deep else N_KINDS_BASE)
        n_deep += deep
        fin = [v for v in vals if v is not None]
        assert o["pert_fail"] == len(vals) - len(fin)
        assert o["spread"] == max(abs(v - o["llh"]) / abs(o["llh"]) for v in fin)
        if o["sens"] is not None:
            assert o["sens"] == max(abs(v - o["llh"]) / abs(o["llh"]) / PERTURB for v in vals[:3])
            assert o["spread"] >= o["sens"] * PERTURB * (1 - 1e-12)
    assert n_deep >= 20


def test_oracle_solver_statistics_match_the_traces():
    """The oracle calls least_squares where the reference does: same number of solves per case as the reference's trace."""
    import gzip
    import json
    import os
    from conftest import GOLDEN
    from scipy import optimize
    traces = {c["name"]: c for c in json.load(gzip.open(os.path.join(GOLDEN, "golden_traces.json.gz"), "rt"))["cases"]}
    by = {c["name"]: c for c in SMALL + SYNTH + SWEEP}
    orig = optimize.least_squares
    for name in ("A4", "A3", "c2_n128_st64_r1", "sw_df_st20_mc9_r0"):
        log = []

        def spy(*a, **kw):
            r = orig(*a, **kw)
            log.append((int(r.nfev), int(r.status)))
            return r
        optimize.least_squares = spy
        try:
            run(by[name])
        finally:
            optimize.least_squares = orig
        want = [(s["nfev"], s["status"]) for s in traces[name]["solves"]]
        assert log == want, name


def test_appendix_a_literals():
    """Known answers quoted in SURVEY.md appendix A."""
    by = {c["name"]: c["out"] for c in SMALL}
    assert by["A1"]["llh"] == -183.18278555708935
    assert by["A3"]["llh"] == -211.9189044185307
    assert by["A7"]["numT"] == 9
    assert by["A1"]["llh_const"] == pytest.approx(4932.701138293154, rel=1e-15)
    assert by["A3"]["llh_const"] == pytest.approx(6926.971304405628, rel=1e-15)


def test_structural_constants():
    """nnz counts quoted in SURVEY.md appendix A."""
    tp = TWO_POP
    nnz = [int((tp.A[0] != 0).sum()), int((tp.A[1] != 0).sum()), int((tp.B[0] != 0).sum()), int((tp.B[1] != 0).sum())]
    M = tp.generator([1.0, 1.0], [1.0, 1.0])
    assert int((M != 0).sum()) == 196
    assert tp.stationary == [30, 31, 34, 35, 38, 39, 42]
    assert int((tp.pulse_matrix(0.3, 0) != 0).sum()) == 132 and int((tp.pulse_matrix(0.3, 1) != 0).sum()) == 132
    for pop in (0, 1):
        np.testing.assert_allclose(tp.pulse_matrix(0.3, pop).sum(axis=0), 1.0, rtol=1e-15)
    assert list(np.diag(ONE_POP.A)) == [-6, -3, -3, -3, -1, -1, -1, -1]
    assert nnz[2] == 88 and nnz[3] == 88
