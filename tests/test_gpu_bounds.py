"""Per-candidate band bounds (misti_eval_batch's band_bounds): the reference's own recommended sweep
(/root/reference/README.md:110-115, `-mi 1 0 {mc} a 0 -mi 2 0 {mc} b 0 -mi 1 {mc} {st} c 0 -mi 2 {mc} {st} d 0
::: st ... ::: mc ...`) evaluated in ONE call, against one reference run per grid point (golden_sweep.json)."""
import numpy as np
import pytest

from conftest import load_golden
from parity import internal_of, llk_bound, spread_of

pytestmark = pytest.mark.gpu

SWEEP = load_golden("golden_sweep")


def sweep_batch(cpfit):
    cases = [c for c in SWEEP if bool(c["in"]["kw"].get("cpfit")) == cpfit]
    split = np.array([c["sweep"]["st"] for c in cases], dtype=float)
    params = np.array([c["sweep"]["rates"] for c in cases], dtype=float)
    bounds = np.array([[[0, c["sweep"]["mc"]], [0, c["sweep"]["mc"]], [c["sweep"]["mc"], -1], [c["sweep"]["mc"], -1]] for c in cases], dtype=np.int32)
    return cases, split, params, bounds


def engine_for(cases, mc=9):
    from misti_amd.engine import Engine
    i = cases[0]["in"]
    bands = [(0, 0, mc, 0.0, 0), (1, 0, mc, 0.0, 1), (0, mc, -1, 0.0, 2), (1, mc, -1, 0.0, 3)]
    kw = i["kw"]
    return Engine(i["times"], i["lambdas"], bands, [], n_param=4, cpfit=bool(kw.get("cpfit")), smooth=bool(kw.get("smooth")),
                  unfolded=bool(kw.get("unfolded")))


@pytest.mark.parametrize("cpfit", [True, False], ids=["cpfit", "default"])
def test_readme_sweep_in_one_call(cpfit):
    cases, split, params, bounds = sweep_batch(cpfit)
    row = [cases[0]["in"]["sfs"]]
    with engine_for(cases) as e:
        r = e.evaluate(split, params, row, band_bounds=bounds)
        shuffled = np.random.default_rng(3).permutation(len(cases))
        r2 = e.evaluate(split[shuffled], params[shuffled], row, band_bounds=bounds[shuffled])
    n_value = 0
    for k, c in enumerate(cases):
        o = c["out"]
        if o["llh"] is None:
            assert r.status[k] == 2 or o["pert_finite"] > 0, (c["name"], r.status[k])
            continue
        if r.status[k] != 0:
            assert o["pert_fail"] > 0 or o.get("internal_fail", 0) > 0, (c["name"], r.status[k])
            continue
        n_value += 1
        bound, clause = llk_bound(o["llh"], c["in"]["sfs"], o["JAFS"], True, spread_of(o), internal_of(o))
        assert abs(r.llk[k, 0] - o["llh"]) <= bound, (c["name"], r.llk[k, 0], o["llh"], bound, clause)
    assert n_value >= 16
    # the order of the candidates in the batch never changes a bit (chains are keyed by parameters AND bounds)
    assert np.array_equal(r2.llk, r.llk[shuffled], equal_nan=True) and np.array_equal(r2.status, r.status[shuffled])
    assert np.array_equal(r2.jafs, r.jafs[shuffled], equal_nan=True)


def test_bounds_equal_one_context_per_boundary():
    """With bounds a grid over {mc} needs one context; without, one context per mc: the results are the same bits."""
    cases, split, params, bounds = sweep_batch(True)
    row = [cases[0]["in"]["sfs"]]
    with engine_for(cases) as e:
        r = e.evaluate(split, params, row, band_bounds=bounds)
    mcs = np.array([c["sweep"]["mc"] for c in cases])
    for mc in sorted(set(mcs)):
        sel = np.where(mcs == mc)[0]
        with engine_for(cases, mc=int(mc)) as e:
            q = e.evaluate(split[sel], params[sel], row)
        assert np.array_equal(q.llk, r.llk[sel], equal_nan=True), mc
        assert np.array_equal(q.jafs, r.jafs[sel], equal_nan=True) and np.array_equal(q.status, r.status[sel])


def test_candidates_differing_only_in_bounds_do_not_share_a_chain():
    """Same split, same rates, different boundary: different likelihoods (the chain key includes the bounds)."""
    cases, split, params, bounds = sweep_batch(True)
    row = [cases[0]["in"]["sfs"]]
    n = 8
    b = np.repeat(bounds[:1], n, axis=0).copy()
    for k in range(n):
        b[k, :2, 1] = 6 + k
        b[k, 2:, 0] = 6 + k
    with engine_for(cases) as e:
        r = e.evaluate(np.full(n, 20.0), np.repeat(params[:1], n, axis=0), row, band_bounds=b)
    assert (r.status == 0).all()
    assert len(set(r.llk[:, 0].tolist())) == n


def test_invalid_bounds_are_a_per_candidate_status():
    """SetModel's checks (MigrationInference.py:237-255) per candidate: status 4 instead of PrintError + exit."""
    cases, split, params, bounds = sweep_batch(True)
    row = [cases[0]["in"]["sfs"]]
    b = np.repeat(bounds[:1], 5, axis=0).copy()
    b[1, 0] = [5, 5]            # start == end
    b[2, 2] = [4, -1]           # overlaps band 0 of the same population ([0, mc))
    b[3, 3] = [25, -1]          # starts after the split (20): start >= end
    b[4, 1] = [0, 40]           # beyond the grid
    with engine_for(cases) as e:
        r = e.evaluate(np.full(5, 20.0), np.repeat(params[:1], 5, axis=0), row, band_bounds=b)
    assert r.status.tolist() == [0, 4, 4, 4, 4]
    assert np.isfinite(r.llk[0, 0]) and np.isneginf(r.llk[1:, 0]).all()
