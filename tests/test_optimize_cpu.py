"""Batched Nelder-Mead reproduces scipy.optimize.minimize(method='Nelder-Mead') start by start."""
import warnings

import numpy as np
import pytest
from scipy import optimize

from misti_amd.optimize import batched_nelder_mead, bootstrap_split_interval, initial_simplex


def rosen_inf(X):
    X = np.atleast_2d(X)
    f = (1 - X[:, 0]) ** 2 + 100 * (X[:, 1] - X[:, 0] ** 2) ** 2
    return np.where((X < 0).any(axis=1), np.inf, f)          # like a negative migration rate: -inf likelihood


def bumpy3(X):
    X = np.atleast_2d(X)
    return np.sum((X - np.array([0.3, 0.05, 0.7])) ** 2, axis=1) + 0.1 * np.sin(20 * X[:, 0]) * np.cos(15 * X[:, 2])


@pytest.mark.parametrize("fun,N,kw", [(rosen_inf, 2, dict(xatol=1e-4, fatol=1e-4, maxiter=1000)),
                                      (bumpy3, 3, dict(xatol=1e-6, fatol=1e-6, maxiter=60)),
                                      (bumpy3, 3, dict(xatol=1e-8, fatol=1e-8))])
def test_matches_scipy_per_start(fun, N, kw):
    rng = np.random.default_rng(1)
    starts = 10.0 ** rng.uniform(-3, 0.3, size=(40, N))
    starts[3, 0] = 0.0                                        # zero coordinate: SciPy's 0.00025 rule
    res = batched_nelder_mead(fun, starts, **kw)
    for i, x0 in enumerate(starts):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            opts = {"xatol": kw["xatol"], "fatol": kw["fatol"]}
            if "maxiter" in kw:
                opts["maxiter"] = kw["maxiter"]
            r = optimize.minimize(lambda x: float(fun(x)[0]), x0, method="Nelder-Mead", options=opts)
        assert np.array_equal(r.x, res.x[i]), (i, r.x, res.x[i])
        assert r.fun == res.fun[i] and r.nit == res.nit[i] and r.nfev == res.nfev[i]


def test_initial_simplex_rule():
    s = initial_simplex(np.array([[1.0, 0.0]]))
    assert np.array_equal(s[0], np.array([[1.0, 0.0], [1.05, 0.0], [1.0, 0.00025]]))


def test_bootstrap_interval():
    rng = np.random.default_rng(0)
    splits = np.arange(40, 60, 0.5)
    llk = -((splits[:, None] - 50.0 - rng.normal(0, 1.5, size=(1, 200))) ** 2)
    mean, (lo, hi), best = bootstrap_split_interval(llk, splits)
    assert best.shape == (200,) and lo < 50.0 < hi and hi - lo < 1.0
