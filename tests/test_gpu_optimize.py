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
