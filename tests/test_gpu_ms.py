"""The TestModel route on the GPU: ms command line -> expected spectrum / likelihood with the true
rates -> forward map (misti_forward_rates), against the reference's own outputs
(tests/golden/golden_ms.json) and, for batches, against the oracle's CoalescentRates."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from parity import JAFS_RTOL, llk_tol

pytestmark = pytest.mark.gpu

CASES = json.load(open(os.path.join(GOLDEN, "golden_ms.json")))["cases"]
IDS = [c["name"] for c in CASES]
FWD_RTOL = 1e-11          # -log(sum exp(M) p / sum p) / T: one 3x3 exponential action per interval


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_testmodel_route_matches_reference(case):
    from misti_amd.io import read_ms
    from misti_amd.engine import MigrationInference
    d = read_ms(case["ms"])
    m = MigrationInference(d.times, d.lambdas, [1] * 8, d.divergenceTime, d.mi, d.pu, unfolded=case["unfolded"], trueEPS=True)
    llh = m.JAFSLikelihood([])
    assert abs(llh - case["llh"]) <= llk_tol(case["llh"], [1] * 8, case["JAFS"], case["unfolded"])
    np.testing.assert_allclose(m.JAFS, case["JAFS"], rtol=JAFS_RTOL)
    np.testing.assert_allclose(np.array(m.lc), np.array(case["lc"]), rtol=1e-12)
    m.CoalescentRates()
    np.testing.assert_allclose(np.array(m.lc), np.array(case["lambdas"]), rtol=0)          # lc := the true rates
    np.testing.assert_allclose(np.array(m.lh), np.array(case["forward_lh"]), rtol=FWD_RTOL)
    np.testing.assert_allclose(np.array(m.Pr), np.array(case["forward_Pr"]), rtol=FWD_RTOL, atol=1e-300)


def test_testmodel_cli(tmp_path):
    from misti_amd import testmodel
    case = CASES[0]
    fout = str(tmp_path / "model.mi")
    out = io.StringIO()
    with contextlib.redirect_stdout(out), contextlib.redirect_stderr(io.StringIO()):
        rc = testmodel.main([case["ms"], "-uf", "-o", fout, "--funits", str(tmp_path / "none.txt")])
    assert rc == 0
    line = [l for l in out.getvalue().splitlines() if l.startswith("Expected SFS")][0]
    got = json.loads(line[len("Expected SFS"):])
    np.testing.assert_allclose(got, case["JAFS"], rtol=JAFS_RTOL)
    rows = [l.split("\t") for l in open(fout).read().splitlines()]
    assert rows[0] == ["#MiSTI2 ver 0.4"]
    rs = [r for r in rows if r[0] == "RS"]
    assert len(rs) == len(case["lambdas"])
    # columns: RS, time, 1/lc1, 1/lc2, 1/lh1, 1/lh2, mi1, mi2, Pr...
    np.testing.assert_allclose([1 / float(r[4]) for r in rs], [l[0] for l in case["forward_lh"]], rtol=FWD_RTOL)
    np.testing.assert_allclose([1 / float(r[2]) for r in rs], [l[0] for l in case["lambdas"]], rtol=1e-15)


def test_forward_batch_matches_oracle():
    """A batch over split x rate (incl. fractional splits, a pulse, both directions) against the oracle."""
    from misti_amd import synth, io as mio
    from misti_amd.engine import Engine
    from oracle.misti_oracle import OracleModel
    inp = mio.merge_psmc(mio.read_psmc_file(io.StringIO(synth.psmc_text(16, 1, synth.THETA_1))),
                         mio.read_psmc_file(io.StringIO(synth.psmc_text(17, 2, synth.THETA_2))))
    numT = len(inp.lambdas)
    bands = [(0, 2, -1, 0.0, 0), (1, 0, 9, 0.4, -1)]
    pulses = [(0, 5, 0.0, 1)]
    rng = np.random.default_rng(5)
    n = 96
    split = rng.integers(10, numT - 2, n).astype(float)
    split[::3] += rng.uniform(0.05, 0.95, len(split[::3]))
    params = np.stack([10 ** rng.uniform(-3, 0.5, n), rng.uniform(0, 0.9, n)], axis=1)
    with Engine(inp.times, inp.lambdas, bands, pulses, n_param=2, true_eps=True) as e:
        lh, pr, status = e.forward_rates(split, params, want_pr=True)
        lh_held, _, _ = e.forward_rates(split, params, hold_mu=True)
    assert (status == 0).all()
    worst = 0.0
    for c in range(n):
        s = split[c]
        end = int(np.ceil(s))
        m = OracleModel(list(inp.times), [list(x) for x in inp.lambdas], [1] * 8, float(s),
                        [[1, 2, end, params[c, 0], 0], [2, 0, 9, 0.4, 0]], [[1, 5, params[c, 1], 0]], trueEPS=True)
        want = np.array(m.coalescent_rates())
        got = lh[c][: m.numT]
        worst = max(worst, np.max(np.abs(got - want) / np.abs(want)))
        np.testing.assert_allclose(got, want, rtol=FWD_RTOL)
        np.testing.assert_allclose(pr[c][: m.splitT + 1].reshape(-1, 3, 2), np.array(m.Pr), rtol=FWD_RTOL, atol=1e-300)
        assert (lh[c][m.numT:] == 0).all()
        m2 = OracleModel(list(inp.times), [list(x) for x in inp.lambdas], [1] * 8, float(s),
                         [[1, 2, end, params[c, 0], 0], [2, 0, 9, 0.4, 0]], [[1, 5, params[c, 1], 0]], trueEPS=True)
        np.testing.assert_allclose(lh_held[c][: m2.numT], np.array(m2.coalescent_rates(hold_mu=True)), rtol=FWD_RTOL)
    print("forward map worst rel", worst)


def test_forward_statuses():
    from misti_amd.engine import Engine
    times = [0.01, 0.02, 0.04, 0.08, 0.16, 0.32, 0.64]
    lh = [[1, 2], [1, 2], [0.8, 1.5], [0.8, 1.5], [1.2, 1.0], [1.2, 1.0], [0.9, 0.9], [0.7, 0.7]]
    with Engine(times, lh, [(0, 1, 5, 0.3, 0)], [], n_param=1, true_eps=True) as e:
        out, _, status = e.forward_rates([5.0, 5.0, -1.0], [[0.3], [-0.1], [0.3]])
    assert list(status) == [0, 1, 4]
    assert np.isfinite(out[0][:8]).all() and np.isnan(out[1]).all() and np.isnan(out[2]).all()
    assert (out[0][5:8] == np.array(lh[5:8])).all()           # rates after the split are returned unchanged
