"""Batched Nelder-Mead on the engine (BASELINE config 3 shape) and the sharded evaluation."""
import contextlib
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg3():
    from misti_amd import workloads
    from misti_amd.engine import Engine, truth_spectrum
    w = workloads.config3(lambda *a: truth_spectrum(*a), n_start=24)
    eng = Engine(w.times, w.lh, **w.engine_kwargs())
    yield w, eng
    eng.close()


def test_batched_search_equals_scipy_per_start(cfg3):
    """Every start's trajectory equals SciPy's Nelder-Mead on the same (GPU) objective, which is what
    MigrationInference.Solve runs (MigrationInference.py:726): same optimum, same iteration count."""
    from scipy import optimize
    from misti_amd.optimize import solve_batched
    w, eng = cfg3
    split = float(w.split_time[0])
    x, llh, r = solve_batched(eng, split, w.params[:6], w.jsfs[0], tol=1e-4, maxiter=1000)
    assert np.isfinite(llh).all()
    for i in range(3):
        def obj(mu):
            if (np.asarray(mu) < 0).any():
                return np.inf
            return -float(eng.evaluate([split], [list(mu)], w.jsfs).llk[0, 0])
        ref = optimize.minimize(obj, w.params[i], method="Nelder-Mead", options={"xatol": 1e-4, "fatol": 1e-4, "maxiter": 1000})
        assert np.array_equal(ref.x, x[i]) and -ref.fun == llh[i] and ref.nit == r.nit[i]
    # the search improves on every start and lands near the truth (rates 0.2 / 0.05) for most
    start_llh = eng.evaluate(np.full(6, split), w.params[:6], w.jsfs).llk[:, 0]
    assert (llh >= start_llh).all()


def test_mirror_solve_uses_same_optimum(cfg3):
    from misti_amd.engine import MigrationInference
    from misti_amd.optimize import solve_batched
    w, eng = cfg3
    split = int(w.split_time[0])
    p0 = w.params[0]
    mi = [[1, 4, split, p0[0], 1], [2, 10, split, p0[1], 1]]
    with contextlib.redirect_stdout(io.StringIO()):
        m = MigrationInference(list(w.times), [list(x) for x in w.lh], list(w.jsfs[0]), split, mi, [],
                               smooth=True, cpfit=True)
        sol = m.Solve(1e-4)
    x, llh, _ = solve_batched(eng, split, [p0], w.jsfs[0], tol=1e-4)
    assert np.array_equal(np.asarray(sol[0]), x[0]) and sol[1] == llh[0]


def test_sharded_evaluation_single_rank(cfg3):
    """evaluate_sharded with the real engine (world size 1 here; ranks > 1 are covered with gloo on CPU)."""
    from misti_amd.dist import evaluate_sharded
    w, eng = cfg3
    full = eng.evaluate(w.split_time, w.params, w.jsfs)
    out = evaluate_sharded(lambda s, p, j: eng.evaluate(s, p, j).llk, w.split_time, w.params, w.jsfs)
    assert np.array_equal(out.numpy(), full.llk, equal_nan=True)


def test_bootstrap_scan_on_device_matches_host_reduction():
    """Config-4-like scan: the [n_split x n_rep] table stays on the device, misti_argmax_dev picks the best split per
    replicate; same answer as the host reduction (numpy argmax) of the host-buffer evaluation."""
    import torch
    from misti_amd import workloads
    from misti_amd.engine import Engine, truth_spectrum
    from misti_amd.optimize import bootstrap_scan_dev, bootstrap_split_interval
    w = workloads.config4(lambda *a: truth_spectrum(*a))
    rows = w.jsfs[:200]
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        host = e.evaluate(w.split_time, w.params, rows)
        mean_h, ci_h, best_h = bootstrap_split_interval(host.llk, w.split_time)
        mean_d, ci_d, best_d = bootstrap_scan_dev(e, w.split_time, rows)
        # the raw reduction incl. candidates without a value and a replicate without any
        llk = torch.tensor([[1.0, -np.inf, np.nan], [3.0, -np.inf, np.nan], [3.0, -np.inf, -np.inf], [np.nan, -np.inf, np.nan]],
                           dtype=torch.float64, device="cuda")
        best = torch.empty(3, dtype=torch.int32, device="cuda")
        val = torch.empty(3, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()                       # the engine issues on its own non-blocking stream
        e.argmax_dev(4, 3, llk.data_ptr(), best.data_ptr(), val.data_ptr())
        e.sync()
    assert np.array_equal(best_h, best_d) and mean_h == mean_d and ci_h == ci_d
    assert best.cpu().tolist() == [1, -1, -1] and val.cpu().tolist()[0] == 3.0


def test_device_resident_search_equals_host_search_and_scipy(cfg3):
    """misti_nm_solve (simplices, values and decisions in HBM) against the host-side batched search and SciPy itself:
    same best vertex, same value, same nit and nfev for every start."""
    from scipy import optimize
    from misti_amd.optimize import solve_batched, solve_batched_dev
    w, eng = cfg3
    split = float(w.split_time[0])
    x_h, llh_h, r_h = solve_batched(eng, split, w.params, w.jsfs[0], tol=1e-4, maxiter=1000)
    x_d, llh_d, r_d = solve_batched_dev(eng, split, w.params, w.jsfs[0], tol=1e-4, maxiter=1000)
    assert np.array_equal(x_d, x_h) and np.array_equal(llh_d, llh_h)
    assert np.array_equal(r_d["nit"], r_h.nit) and np.array_equal(r_d["nfev"], r_h.nfev)
    assert (r_d["status"] == 0).all() == bool(r_h.converged.all())
    for i in (0, 7, 23):
        def obj(mu):
            if (np.asarray(mu) < 0).any():
                return np.inf
            return -float(eng.evaluate([split], [list(mu)], w.jsfs).llk[0, 0])
        ref = optimize.minimize(obj, w.params[i], method="Nelder-Mead", options={"xatol": 1e-4, "fatol": 1e-4, "maxiter": 1000})
        assert np.array_equal(ref.x, x_d[i]) and -ref.fun == llh_d[i] and ref.nit == r_d["nit"][i] and ref.nfev == r_d["nfev"][i]


def test_device_search_iteration_budget_and_fractional_split(cfg3):
    """maxiter as SciPy counts it (iterations start at 1: at most maxiter - 1 are run), on a fractional split time."""
    from scipy import optimize
    from misti_amd.optimize import solve_batched_dev
    w, eng = cfg3
    split = float(w.split_time[0]) - 0.5
    x, llh, r = solve_batched_dev(eng, split, w.params[:4], w.jsfs[0], tol=1e-9, maxiter=12)
    assert (r["status"] == 2).all() and (r["nit"] == 12).all()
    def obj(mu):
        if (np.asarray(mu) < 0).any():
            return np.inf
        return -float(eng.evaluate([split], [list(mu)], w.jsfs).llk[0, 0])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = optimize.minimize(obj, w.params[2], method="Nelder-Mead", options={"xatol": 1e-9, "fatol": 1e-9, "maxiter": 12})
    assert np.array_equal(ref.x, x[2]) and -ref.fun == llh[2] and ref.nfev == r["nfev"][2]


def test_device_search_at_config3_size():
    """BASELINE config 3: 16 384 random starts, two optimised bands, searched in one call with the simplices in HBM;
    32 sampled starts bit-equal to scipy.optimize.minimize(method='Nelder-Mead') on the GPU objective."""
    from scipy import optimize
    from misti_amd import workloads
    from misti_amd.engine import Engine, truth_spectrum
    from misti_amd.optimize import solve_batched_dev
    w = workloads.config3(lambda *a: truth_spectrum(*a))
    assert w.n_cand == 16384
    split = float(w.split_time[0])
    with Engine(w.times, w.lh, **w.engine_kwargs()) as eng:
        x, llh, r = solve_batched_dev(eng, split, w.params, w.jsfs[0], tol=1e-4, maxiter=1000)
        assert np.isfinite(llh).mean() > 0.99
        assert (r["status"] == 0).mean() > 0.99
        start_llh = eng.evaluate(np.full(w.n_cand, split), w.params, w.jsfs).llk[:, 0]
        ok = np.isfinite(start_llh)
        assert (llh[ok] >= start_llh[ok]).all()
        rng = np.random.default_rng(11)
        for i in rng.choice(w.n_cand, 32, replace=False):
            def obj(mu):
                if (np.asarray(mu) < 0).any():
                    return np.inf
                v = float(eng.evaluate([split], [list(mu)], w.jsfs).llk[0, 0])
                return -v if np.isfinite(v) else np.inf
            ref = optimize.minimize(obj, w.params[i], method="Nelder-Mead", options={"xatol": 1e-4, "fatol": 1e-4, "maxiter": 1000})
            assert np.array_equal(ref.x, x[i]) and -ref.fun == llh[i], (i, ref.x, x[i])
            assert ref.nit == r["nit"][i] and ref.nfev == r["nfev"][i]
    # the surface has one dominant basin: most starts end within the tolerance of the best value found
    assert (llh >= np.max(llh) - 1e-3).mean() > 0.8


def test_speculative_iterations_do_not_change_the_search(cfg3):
    """With few starts left an iteration sends all 4 + N points SciPy could ask for as ONE batch (misti_nm.hip:
    nm_spec_points / nm_spec_finish); decisions and counters must be those of the three-batch path, bit for bit."""
    import os
    from misti_amd.engine import Engine
    w, eng = cfg3
    split = float(w.split_time[0])
    out = {}
    for name, env in (("spec", {}), ("plain", {"MISTI_NM_SPEC": "0"})):
        os.environ.update(env)
        try:
            with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
                out[name] = e.nm_solve(w.params[:300], split, w.jsfs[0], tol=1e-4, maxiter=1000)
        finally:
            for k in env:
                os.environ.pop(k, None)
    a, b = out["spec"], out["plain"]
    assert a["speculative_iterations"] > 10 and b["speculative_iterations"] == 0
    for k in ("x", "llh", "nit", "nfev", "status"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k


def test_basinhopping_equals_scipy(cfg3):
    """The reference's global search (MigrationInference.Solve(globalOpt=True), /root/reference/MigrationInference.py:723-725:
    scipy.optimize.basinhopping(T=0.5, Nelder-Mead)) for 8 starts at once on the device against SciPy's own runner on the GPU
    objective, one numpy Generator per start: lowest minimum, its value, the evaluation count and the failed minimisations."""
    from scipy import optimize
    w, eng = cfg3
    split = float(w.split_time[0])
    S, niter = 8, 7
    starts = w.params[5:5 + S]
    got = eng.basinhopping(starts, split, w.jsfs[0], rngs=[100 + s for s in range(S)], niter=niter, T=0.5, stepsize=0.05, interval=3)

    def obj(mu):
        if (np.asarray(mu) < 0).any():
            return np.inf
        v = float(eng.evaluate([split], [list(mu)], w.jsfs).llk[0, 0])
        return -v if np.isfinite(v) else np.inf
    import warnings
    hops_taken = 0
    for s in range(S):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = optimize.basinhopping(obj, starts[s], niter=niter, T=0.5, stepsize=0.05, interval=3,
                                        minimizer_kwargs=dict(method="Nelder-Mead"), rng=np.random.default_rng(100 + s))
        assert np.array_equal(ref.x, got["x"][s]), (s, ref.x, got["x"][s])
        assert -ref.fun == got["llh"][s] and ref.nfev == got["nfev"][s] and ref.minimization_failures == got["failures"][s], (s, ref.fun, ref.nfev)
        hops_taken += got["accepted"][s]
    assert hops_taken > 0                                    # the Metropolis test really accepted some hops


def test_evaluation_budget_is_cut_per_evaluation_as_scipy_does(cfg3):
    """maxfev: SciPy refuses the call that would exceed the budget and abandons the iteration in progress (nothing accepted, a shrink
    applied up to the refused vertex, `iterations` not incremented: _minimize_neldermead's _MaxFuncCallError path).  Budgets 3 ... 25
    cut the minimisation in every phase - inside the initial simplex, before the second point, inside a shrink; compared with
    scipy.optimize.basinhopping(niter=0 and 3) on the GPU objective, whose result carries the minimisation's x, fun, nfev and failure."""
    from scipy import optimize
    import warnings
    w, eng = cfg3
    split = float(w.split_time[0])
    S = 6
    starts = w.params[:S]

    def obj(mu):
        if (np.asarray(mu) < 0).any():
            return np.inf
        v = float(eng.evaluate([split], [list(mu)], w.jsfs).llk[0, 0])
        return -v if np.isfinite(v) else np.inf
    cut_seen = 0
    for maxfev, niter in ((2, 0), (3, 0), (4, 0), (5, 0), (6, 0), (7, 0), (9, 0), (12, 0), (16, 0), (25, 0), (8, 3), (13, 3)):
        got = eng.basinhopping(starts, split, w.jsfs[0], rngs=[300 + s for s in range(S)], niter=niter, T=0.5, stepsize=0.05, interval=2,
                               nm_maxiter=10 ** 6, nm_maxfev=maxfev)
        for s in range(S if niter == 0 else 3):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref = optimize.basinhopping(obj, starts[s], niter=niter, T=0.5, stepsize=0.05, interval=2,
                                            minimizer_kwargs=dict(method="Nelder-Mead", options=dict(maxfev=maxfev, maxiter=10 ** 6)),
                                            rng=np.random.default_rng(300 + s))
            assert np.array_equal(ref.x, got["x"][s]), (maxfev, niter, s, ref.x, got["x"][s])
            assert -ref.fun == got["llh"][s] and ref.nfev == got["nfev"][s], (maxfev, niter, s, ref.fun, got["llh"][s], ref.nfev, got["nfev"][s])
            assert ref.minimization_failures == got["failures"][s]
            assert got["nfev"][s] <= (niter + 1) * maxfev
            cut_seen += got["failures"][s] > 0
    assert cut_seen >= 20                                   # the budget really ran out


def test_basinhopping_at_config3_size():
    """BASELINE config 3 as worded - basin hopping from 16 384 random starts - in one call (2 hops per start here; bench.py
    --workload config3-basinhopping runs SciPy's default of 100, /root/reference/MigrationInference.py:723-725); 8 sampled starts
    bit-equal to scipy.optimize.basinhopping on the GPU objective, each with the generator the device was given for that start."""
    from scipy import optimize
    import warnings
    from misti_amd import workloads
    from misti_amd.engine import Engine, truth_spectrum
    w = workloads.config3(lambda *a: truth_spectrum(*a))
    assert w.n_cand == 16384
    split = float(w.split_time[0])
    niter = 2
    with Engine(w.times, w.lh, **w.engine_kwargs()) as eng:
        got = eng.basinhopping(w.params, split, w.jsfs[0], rngs=list(5000 + np.arange(w.n_cand)), niter=niter, T=0.5, stepsize=0.5)
        assert np.isfinite(got["llh"]).mean() > 0.99
        assert (got["nfev"] <= (niter + 1) * 400).all()          # three minimisations, each within SciPy's default budget of 200 x N
        assert (got["failures"] <= niter + 1).all() and got["accepted"].max() <= niter

        def obj(mu):
            if (np.asarray(mu) < 0).any():
                return np.inf
            v = float(eng.evaluate([split], [list(mu)], w.jsfs).llk[0, 0])
            return -v if np.isfinite(v) else np.inf
        rng = np.random.default_rng(17)
        sample = list(rng.choice(w.n_cand, 7, replace=False)) + [int(np.argmax(got["nfev"]))]      # ... and the start that spent the most evaluations
        for s in sample:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref = optimize.basinhopping(obj, w.params[s], niter=niter, T=0.5, stepsize=0.5, minimizer_kwargs=dict(method="Nelder-Mead"),
                                            rng=np.random.default_rng(5000 + int(s)))
            assert np.array_equal(ref.x, got["x"][s]), (s, ref.x, got["x"][s])
            assert -ref.fun == got["llh"][s] and ref.nfev == got["nfev"][s] and ref.minimization_failures == got["failures"][s], (s, ref.nfev, got["nfev"][s])
