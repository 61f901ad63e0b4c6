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
