"""Full-size parity of the HIP path through the C ABI: the config-2 grid against the
oracle on a sample, and size-independent properties of the model on the whole batch."""
import numpy as np
import pytest

from parity import pinned, baseline_contract, llk_tol, record

pytestmark = pytest.mark.gpu

RUNAWAY = 5.0       # oracle's max corrected rate x interval length above which the reference is noise-driven
REGULAR_BEYOND_MEASURED = 0


@pytest.fixture(scope="module")
def cfg2():
    from misti_amd import workloads
    from misti_amd.engine import Engine, truth_spectrum
    w = workloads.config2(lambda *a: truth_spectrum(*a))
    eng = Engine(w.times, w.lh, **w.engine_kwargs())
    res = eng.evaluate(w.split_time, w.params, w.jsfs)
    yield w, eng, res
    eng.close()


def test_grid_sample_against_oracle(cfg2):
    """BASELINE config 2 at full size (4 096 candidates, numT = 128) vs the oracle on 96
    evenly spaced candidates: statuses equal; llk within 1e-9 (+ rounding floor) wherever the
    reference is determined; runaway-rate candidates under the per-candidate contract of tests/parity.py."""
    from oracle.batch import oracle_batch
    w, eng, res = cfg2
    idx = np.linspace(0, w.n_cand - 1, 96).astype(int)
    o_llk, o_st, _ = oracle_batch(w, idx, processes=8)
    run = oracle_batch.last_runaway
    assert (res.status[idx] == o_st).all()
    n_reg = n_out = 0
    for k, c in enumerate(idx):
        if o_st[k] != 0:
            continue
        err = abs(res.llk[c, 0] - o_llk[k, 0])
        if run[k] < RUNAWAY:
            n_reg += 1
            # a stop/continue flip of SciPy's gtol test (|J^T f| within rounding of 1e-10) moves llk by up to a few
            # 1e-9; observed on ~0.1 % of regular candidates.  Everything else meets the 1e-9 contract.
            if err > llk_tol(o_llk[k, 0], w.jsfs[0], res.jafs[c], False):
                n_out += 1                      # held to the per-candidate contract below, like every other candidate of the sample
            else:
                np.testing.assert_allclose(res.jafs[c], oracle_batch.last_jafs[k], rtol=1e-9)
    record("test_grid_sample_against_oracle", regular=n_reg, regular_beyond_1e9=n_out)
    # measured on MI355X (profiles/r04_measured_guards.jsonl): 80 regular candidates in the sample, NONE beyond 1e-9; the guard is measured + 1
    assert n_reg >= 60
    pinned(n_out <= REGULAR_BEYOND_MEASURED + 1, ("regular candidates beyond 1e-9", n_out))                    # each is held to the per-candidate contract below
    # runaway-rate candidates: the per-candidate contract (1e-9, or 10 x that candidate's own spread under eight 2^-48
    # perturbations, measured here through the compiled baseline).  Measured on MI355X: 85 tight, 11 within their spread, none outside
    rep = baseline_contract(w, idx, res.llk, res.status)
    record("test_grid_sample_contract", tight=rep["tight"], self_bound=rep["self_bound"], outside=[int(idx[k]) for k in rep["outside"]], worst_factor=float(rep["factor"].max()))
    assert len(rep["mismatch"]) == 0 and len(rep["outside"]) == 0, [(int(idx[k]), float(rep["rel"][k]), float(rep["factor"][k])) for k in rep["outside"]]
    # the engine's own diagnostic agrees with the oracle's measure of the same quantity
    ok = (o_st == 0)
    both_small = (run[ok] < 4.0) & (res.runaway[idx][ok] < 4.0)
    both_big = (run[ok] > 6.0) & (res.runaway[idx][ok] > 6.0)
    assert (both_small | both_big | ((run[ok] >= 4.0) & (run[ok] <= 6.0))).mean() > 0.98


def test_spectrum_is_a_distribution(cfg2):
    w, eng, res = cfg2
    ok = res.status == 0
    assert ok.mean() > 0.9
    j = res.jafs[ok]
    assert (j > 0).all()
    np.testing.assert_allclose(j.sum(axis=1), 1.0, rtol=1e-14)
    assert np.isneginf(res.llk[~ok]).all() and np.isnan(res.jafs[~ok]).all()


def test_deterministic_and_batch_independent(cfg2):
    """Same inputs -> bitwise the same outputs; a candidate evaluated alone or in any batch
    position gives bitwise the same result (no cross-candidate state)."""
    w, eng, res = cfg2
    again = eng.evaluate(w.split_time, w.params, w.jsfs)
    assert np.array_equal(again.llk, res.llk) and np.array_equal(again.status, res.status)
    pick = np.array([0, 17, 1234, 2047, 4095, 3000, 5])
    sub = eng.evaluate(w.split_time[pick], w.params[pick], w.jsfs)
    assert np.array_equal(sub.llk, res.llk[pick])
    assert np.array_equal(sub.jafs, res.jafs[pick], equal_nan=True)


def test_time_scale_invariance(cfg2):
    """Rescaling time (times * a, every rate / a) leaves the normalised spectrum unchanged."""
    from misti_amd.engine import Engine
    w, eng, res = cfg2
    a = 4.0                                   # a power of two: the rescaled inputs are exact
    kw = w.engine_kwargs()
    with Engine(np.array(w.times) * a, np.array(w.lh) / a, **kw) as e2:
        r2 = e2.evaluate(w.split_time, w.params / a, w.jsfs)
    ok = (res.status == 0) & (r2.status == 0)
    assert (res.status == r2.status).mean() > 0.99
    d = np.full(w.n_cand, 0.0)
    d[ok] = np.abs(r2.jafs[ok] / res.jafs[ok] - 1).max(axis=1)
    assert np.quantile(d[ok], 0.8) < 1e-12      # exact up to rounding for regular candidates
    # The others are runaway-rate candidates, whose stopping point is noise-driven in the reference itself: no invariance to claim, but
    # the evaluation on the RESCALED inputs must meet the per-candidate contract on those inputs (1e-9, or SELF_FACTOR x that candidate's
    # own spread, through the compiled baseline) - no blanket tolerance.
    moved = np.where(ok & (d >= 1e-9))[0]
    assert (res.runaway[moved] >= RUNAWAY).all()
    if len(moved):
        from types import SimpleNamespace
        w2 = SimpleNamespace(**{**w.__dict__, "times": list(np.array(w.times) * a), "lh": [list(x) for x in np.array(w.lh) / a], "params": w.params / a})
        rep = baseline_contract(w2, moved, r2.llk, r2.status)
        assert len(rep["mismatch"]) == 0 and len(rep["outside"]) == 0, [(int(moved[k]), float(rep["rel"][k]), float(rep["factor"][k])) for k in rep["outside"]]


def test_population_swap_symmetry(cfg2):
    """Swapping the two genomes/populations permutes the JSFS classes
    (0100<->0001, 1100<->0011, 1101<->0111; 0101 fixed)."""
    from misti_amd.engine import Engine
    w, eng, res = cfg2
    lh = np.array(w.lh)[:, ::-1].copy()
    bands = [(1 - p, s, e, v, par) for p, s, e, v, par in w.bands]
    kw = w.engine_kwargs()
    kw["bands"] = bands
    with Engine(w.times, lh, **kw) as e2:
        r2 = e2.evaluate(w.split_time, w.params, w.jsfs)
    perm = [2, 5, 0, 3, 6, 1, 4]
    ok = (res.status == 0) & (r2.status == 0)
    d = np.full(w.n_cand, 0.0)
    d[ok] = np.abs(r2.jafs[ok][:, perm] / res.jafs[ok] - 1).max(axis=1)
    assert np.quantile(d[ok], 0.8) < 1e-11
    # runaway-rate candidates: the swapped evaluation under the per-candidate contract on the swapped inputs (see test_time_scale_invariance)
    moved = np.where(ok & (d >= 1e-9))[0]
    assert (res.runaway[moved] >= RUNAWAY).all()
    if len(moved):
        from types import SimpleNamespace
        w2 = SimpleNamespace(**{**w.__dict__, "lh": [list(x) for x in lh], "bands": bands})
        rep = baseline_contract(w2, moved, r2.llk, r2.status)
        assert len(rep["mismatch"]) == 0 and len(rep["outside"]) == 0, [(int(moved[k]), float(rep["rel"][k]), float(rep["factor"][k])) for k in rep["outside"]]


def test_replicate_linearity(cfg2):
    """llk(c, r) - llh_const(r) is linear in the data row (multinomial log-likelihood):
    checked for a + b against a and b on every candidate of the grid."""
    import math
    w, eng, res = cfg2
    a = w.jsfs[0]
    b = np.array([a[0], 7, 1, 9, 3, 2, 8, 5], dtype=float) * 1000.0
    rows = np.stack([a, b, a + b])
    r = eng.evaluate(w.split_time, w.params, rows)

    def const(row):
        d = row[1:]
        f = [d[0] + d[6], d[1] + d[5], d[2] + d[4], d[3]]
        return math.lgamma(sum(d) + 1) - sum(math.lgamma(v + 1) for v in f)
    ok = r.status == 0
    core = r.llk[ok] - np.array([const(x) for x in rows])[None, :]
    np.testing.assert_allclose(core[:, 2], core[:, 0] + core[:, 1], rtol=1e-9)


def test_edge_cases():
    """Empty batch, no replicates, fractional split at full size, bad candidates."""
    from misti_amd import workloads
    from misti_amd.engine import Engine, truth_spectrum
    w = workloads.config4(lambda *a: truth_spectrum(*a), n_split=16, n_rep=8, cpfit=True)
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        r0 = e.evaluate(np.zeros(0), None, w.jsfs)
        assert r0.llk.shape == (0, 8)
        r1 = e.evaluate(w.split_time, None, None)
        assert r1.llk.shape == (16, 0) and (r1.status == 0).all()
        r2 = e.evaluate(w.split_time, None, w.jsfs)
        assert r2.llk.shape == (16, 8) and np.isfinite(r2.llk).all()
        bad = e.evaluate([-1.0, 500.0, 127.5, 128.0], None, w.jsfs)
        assert list(bad.status) == [4, 4, 4, 3] and np.isneginf(bad.llk).all()
