"""The TestModel route on the CPU: the ms reader against the reference reader's dumps and the
oracle's forward map against the reference's CoalescentRates (tests/golden/golden_ms.json,
written by tests/golden/make_golden.py from the reference itself)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

CASES = json.load(open(os.path.join(GOLDEN, "golden_ms.json")))["cases"]
IDS = [c["name"] for c in CASES]


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_read_ms_matches_reference_reader(case):
    from misti_amd.io import read_ms
    d = read_ms(case["ms"])
    assert d.times == case["times"]                          # exact: same arithmetic on the same literals
    assert d.lambdas == case["lambdas"]
    assert d.divergenceTime == case["divergenceTime"]
    assert [list(map(float, m)) for m in d.mi] == case["mi"]
    assert [list(map(float, q)) for q in d.pu] == case["pu"]
    assert d.scaleTime == 1.0 and d.theta == 1.0


def test_read_ms_rejects_what_the_reference_rejects(capsys):
    from misti_amd.io import read_ms
    with pytest.raises(SystemExit) as e:                     # no -ej: "Populations should be merged"
        read_ms("4 1 -t 10 -I 2 2 2 -n 1 2.0")
    assert e.value.code == 0
    assert "Populations should be merged" in capsys.readouterr().out
    with pytest.raises(SystemExit):
        read_ms("4 1 -I 2 2 2 -n 3 2.0 -ej 0.1 2 1")


def test_anchor_value_of_the_survey():
    """the TestModel-route anchor."""
    c = CASES[0]
    assert c["llh"] == -6.0352508984328725
    assert c["JAFS"][0] == 0.2286622684039319 and c["JAFS"][3] == 0.17318429024933293


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_on_the_testmodel_route(case):
    from oracle.misti_oracle import OracleModel
    m = OracleModel(list(case["times"]), [list(x) for x in case["lambdas"]], [1] * 8, case["divergenceTime"],
                    [list(x) for x in case["mi"]], [list(x) for x in case["pu"]], unfolded=case["unfolded"], trueEPS=True)
    llh = m.jafs_likelihood([])
    assert llh == pytest.approx(case["llh"], rel=1e-13)
    np.testing.assert_allclose(m.JAFS, case["JAFS"], rtol=1e-13)
    lh = m.coalescent_rates(hold_mu=True)            # the reference's behaviour (mu left over from the likelihood call)
    np.testing.assert_allclose(np.array(lh), np.array(case["forward_lh"]), rtol=1e-13)
    np.testing.assert_allclose(np.array(m.Pr), np.array(case["forward_Pr"]), rtol=1e-13, atol=1e-300)


@pytest.mark.parametrize("case", CASES[1:3], ids=IDS[1:3])
def test_per_interval_forward_map_against_independent_host_code(case):
    """hold_mu=False (each interval's own migration rates) has no reference output to compare with
    (the reference cannot do it); pin the oracle's version against misti_amd.synth.forward_rates,
    written independently (own 3x3 exponential)."""
    from oracle.misti_oracle import OracleModel
    from misti_amd import synth
    m = OracleModel(list(case["times"]), [list(x) for x in case["lambdas"]], [1] * 8, case["divergenceTime"],
                    [list(x) for x in case["mi"]], [list(x) for x in case["pu"]], unfolded=case["unfolded"], trueEPS=True)
    mi, pu = [list(r) for r in m.mi], [list(r) for r in m.pu]
    want = synth.forward_rates(case["times"], case["lambdas"], case["divergenceTime"], mi, pu)
    got = m.coalescent_rates()
    np.testing.assert_allclose(np.array(got), np.array(want), rtol=1e-12)
    assert np.max(np.abs(np.array(got) - np.array(case["forward_lh"]))) > 1e-3      # and it differs from the held-mu variant
