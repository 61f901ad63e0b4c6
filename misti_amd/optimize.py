"""Batched Nelder-Mead: many independent simplices advanced together, every round of
objective evaluations issued as ONE batch to the likelihood engine.

The reference optimises one model at a time with SciPy's Nelder-Mead
(``MigrationInference.Solve``, ``/root/reference/MigrationInference.py:718-733``:
``method='Nelder-Mead', xatol = fatol = tol, maxiter = 1000``, started from the
``-mi``/``-pu`` initial values; a basin-hopping variant exists but is unreachable from
its CLI).  BASELINE config 3 runs that search from 16 384 random starts.  Each start
here follows exactly SciPy's (non-adaptive) iteration - same initial simplex, same
reflection / expansion / contraction / shrink decisions, same termination test - so a
start's trajectory equals ``scipy.optimize.minimize(..., method='Nelder-Mead')`` on the
same objective; only the evaluation is batched: per iteration at most three engine
calls (reflection points of all live starts; their expansion/contraction points; the
shrunk vertices of those that shrink).
"""
from __future__ import annotations

import numpy as np

RHO, CHI, PSI, SIGMA = 1.0, 2.0, 0.5, 0.5          # scipy/optimize/_optimize.py, adaptive=False
NONZDELT, ZDELT = 0.05, 0.00025


class NMResult:
    __slots__ = ("x", "fun", "nit", "nfev", "converged", "simplex", "fsim")

    def __init__(self, x, fun, nit, nfev, converged, simplex, fsim):
        self.x, self.fun, self.nit, self.nfev, self.converged = x, fun, nit, nfev, converged
        self.simplex, self.fsim = simplex, fsim


def initial_simplex(x0):
    """[S, N] starts -> [S, N+1, N] simplices as SciPy builds them."""
    x0 = np.asarray(x0, dtype=float)
    S, N = x0.shape
    sim = np.repeat(x0[:, None, :], N + 1, axis=1)
    for k in range(N):
        y = x0[:, k]
        sim[:, k + 1, k] = np.where(y != 0, (1 + NONZDELT) * y, ZDELT)
    return sim


def _sort(sim, fsim):
    order = np.argsort(fsim, axis=1)          # as SciPy: default argsort (insertion sort at these sizes)
    return np.take_along_axis(sim, order[:, :, None], axis=1), np.take_along_axis(fsim, order, axis=1)


def batched_nelder_mead(fun_batch, x0, xatol=1e-4, fatol=1e-4, maxiter=None, maxfun=None):
    """Minimise ``fun`` from every row of ``x0``.

    ``fun_batch(X[M, N]) -> f[M]`` evaluates M points at once (``+inf`` allowed).
    Returns an ``NMResult`` of arrays over the S starts.
    """
    x0 = np.atleast_2d(np.asarray(x0, dtype=float))
    S, N = x0.shape
    if maxiter is None and maxfun is None:              # SciPy's defaults (_minimize_neldermead)
        maxiter = maxfun = N * 200
    elif maxiter is None:
        maxiter = N * 200 if maxfun == np.inf else np.inf
    elif maxfun is None:
        maxfun = N * 200 if maxiter == np.inf else np.inf
    sim = initial_simplex(x0)
    fsim = np.asarray(fun_batch(sim.reshape(S * (N + 1), N)), dtype=float).reshape(S, N + 1)
    nfev = np.full(S, N + 1)
    sim, fsim = _sort(sim, fsim)
    nit = np.ones(S, dtype=int)                 # SciPy starts its iteration counter at 1
    live = np.ones(S, dtype=bool)

    def done_mask():
        dx = np.max(np.abs(sim[:, 1:, :] - sim[:, :1, :]), axis=(1, 2))
        with np.errstate(invalid="ignore"):
            df = np.max(np.abs(fsim[:, :1] - fsim[:, 1:]), axis=1)
        return (dx <= xatol) & (df <= fatol)

    while True:
        live &= ~done_mask()
        live &= (nit < maxiter) & (nfev < maxfun)
        idx = np.where(live)[0]
        if idx.size == 0:
            break
        xbar = sim[idx, :-1, :].sum(axis=1) / N
        worst = sim[idx, -1, :]
        xr = (1 + RHO) * xbar - RHO * worst
        fxr = np.asarray(fun_batch(xr), dtype=float)
        nfev[idx] += 1
        f0, fn1, fn = fsim[idx, 0], fsim[idx, -2], fsim[idx, -1]
        want_e = fxr < f0
        mid = ~want_e & (fxr < fn1)
        want_c = ~want_e & ~mid & (fxr < fn)
        want_cc = ~want_e & ~mid & ~want_c
        # second round: one extra point for everything but the plain reflections
        x2 = np.where(want_e[:, None], (1 + RHO * CHI) * xbar - RHO * CHI * worst,
                      np.where(want_c[:, None], (1 + PSI * RHO) * xbar - PSI * RHO * worst,
                               (1 - PSI) * xbar + PSI * worst))
        need2 = ~mid
        f2 = np.full(idx.size, np.nan)
        if need2.any():
            f2[need2] = np.asarray(fun_batch(x2[need2]), dtype=float)
            nfev[idx[need2]] += 1
        new_x = xr.copy()
        new_f = fxr.copy()
        shrink = np.zeros(idx.size, dtype=bool)
        take_e = want_e & (f2 < fxr)
        new_x[take_e], new_f[take_e] = x2[take_e], f2[take_e]
        ok_c = want_c & (f2 <= fxr)
        new_x[ok_c], new_f[ok_c] = x2[ok_c], f2[ok_c]
        shrink |= want_c & ~ok_c
        ok_cc = want_cc & (f2 < fn)
        new_x[ok_cc], new_f[ok_cc] = x2[ok_cc], f2[ok_cc]
        shrink |= want_cc & ~ok_cc
        keep = ~shrink
        sim[idx[keep], -1, :] = new_x[keep]
        fsim[idx[keep], -1] = new_f[keep]
        if shrink.any():
            si = idx[shrink]
            sim[si, 1:, :] = sim[si, :1, :] + SIGMA * (sim[si, 1:, :] - sim[si, :1, :])
            fs = np.asarray(fun_batch(sim[si, 1:, :].reshape(si.size * N, N)), dtype=float).reshape(si.size, N)
            fsim[si, 1:] = fs
            nfev[si] += N
        s2, f2s = _sort(sim[idx], fsim[idx])
        sim[idx], fsim[idx] = s2, f2s
        nit[idx] += 1
    return NMResult(sim[:, 0, :].copy(), fsim[:, 0].copy(), nit, nfev, done_mask(), sim, fsim)


def solve_batched(engine, split_time, starts, jsfs_row, tol=1e-4, maxiter=1000):
    """``MigrationInference.Solve`` for many starts: maximise the likelihood of one data JSFS
    over the optimised band/pulse parameters from each row of ``starts`` ([S, P]).

    ``engine`` is a ``misti_amd.engine.Engine``; negative parameters give ``-inf`` exactly as
    the reference's guard (MigrationInference.py:569-572).  Returns (params[S, P], llh[S], NMResult).
    """
    starts = np.atleast_2d(np.asarray(starts, dtype=float))
    row = np.asarray(jsfs_row, dtype=float).reshape(1, 8)

    def objective(X):
        res = engine.evaluate(np.full(X.shape[0], float(split_time)), X, row)
        return -res.llk[:, 0]

    r = batched_nelder_mead(objective, starts, xatol=tol, fatol=tol, maxiter=maxiter)
    return r.x, -r.fun, r


def _world(group=None):
    """Ranks of the process group this process belongs to (1 without torch.distributed or before it is initialised)."""
    try:
        import torch.distributed as dist
    except ImportError:
        return 1
    return dist.get_world_size(group) if dist.is_initialized() else 1


def solve_batched_dev(engine, split_time, starts, jsfs_row, tol=1e-4, maxiter=1000, group=None):
    """``solve_batched`` with the simplices resident on the device (``misti_nm_solve``): no host round trip per
    iteration - per iteration three engine batches and four one-thread-per-start kernels on the engine's stream.
    Same decisions as SciPy's Nelder-Mead, hence the same result as ``solve_batched`` / ``MigrationInference.Solve``.
    Inside a process group (one rank per GPU: ``misti_amd.dist.init_from_env``) the starts are dealt to the ranks in contiguous
    blocks, every rank searches its block on its own GPU and one all_gather returns all starts on every rank
    (``dist.search_sharded``) - BASELINE config 3's 16 384 starts are 2 048 per GPU on a node; results are those of one device.
    Returns (params[S, P], llh[S], dict with nit, nfev, status)."""
    if _world(group) > 1:
        from . import dist as mdist
        r = mdist.search_sharded(lambda st: engine.nm_solve(st, split_time, jsfs_row, tol=tol, maxiter=maxiter), starts,
                                 ("x", "llh", "nit", "nfev", "status"), group=group)
        return r["x"], r["llh"], r
    r = engine.nm_solve(starts, split_time, jsfs_row, tol=tol, maxiter=maxiter)
    return r["x"], r["llh"], r


def basinhopping_dev(engine, split_time, starts, jsfs_row, rngs, group=None, **kw):
    """The reference's global search (``MigrationInference.Solve(globalOpt=True)``, ``/root/reference/MigrationInference.py:723-725``)
    from every row of ``starts`` (``Engine.basinhopping``: SciPy's runner step for step); inside a process group the starts - and
    their generators ``rngs`` - are dealt to the ranks in contiguous blocks and gathered once (``dist.search_sharded``).
    Returns dict(x, llh, nfev, failures, accepted)."""
    if _world(group) > 1:
        from . import dist as mdist
        return mdist.search_sharded(lambda st, rngs: engine.basinhopping(st, split_time, jsfs_row, rngs, **kw), starts,
                                    ("x", "llh", "nfev", "failures", "accepted"), group=group, rngs=list(rngs))
    return engine.basinhopping(starts, split_time, jsfs_row, rngs, **kw)


def solve_grouped_dev(engines, split_time, starts, jsfs_row, tol=1e-4, maxiter=1000):
    """``solve_batched_dev`` with the starts dealt out to several engine contexts (one host thread each; ctypes releases the
    GIL for the duration of ``misti_nm_solve``): the three engine batches of an iteration depend on each other, the searches of
    different groups do not, so their batches overlap on the GPU.  A start's trajectory does not depend on which batch its
    points travel in: results equal ``solve_batched_dev`` on one context, bit for bit.
    Returns (params[S, P], llh[S], dict with nit, nfev, status and per-group work counters)."""
    import threading
    starts = np.atleast_2d(np.asarray(starts, dtype=float))
    G = len(engines)
    parts = [np.arange(g, starts.shape[0], G) for g in range(G)]              # interleaved: every group sees the same mix of starts
    res, err = [None] * G, []

    def work(g):
        try:
            res[g] = engines[g].nm_solve(starts[parts[g]], split_time, jsfs_row, tol=tol, maxiter=maxiter)
        except BaseException as e:                                           # surfaces in the caller's thread
            err.append(e)
    threads = [threading.Thread(target=work, args=(g,)) for g in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if err:
        raise err[0]
    S, P = starts.shape
    out = dict(x=np.empty((S, P)), llh=np.empty(S), nit=np.empty(S, dtype=np.int32), nfev=np.empty(S, dtype=np.int32), status=np.empty(S, dtype=np.int32))
    for g in range(G):
        for k in out:
            out[k][parts[g]] = res[g][k]
    out["iterations_issued"] = max(r["iterations_issued"] for r in res)
    out["slots"] = sum(r["slots"] for r in res)
    out["speculative_iterations"] = max(r["speculative_iterations"] for r in res)
    out["groups"] = G
    return out["x"], out["llh"], out


def bootstrap_split_interval(llk, split_values, level=0.95):
    """Confidence interval of the split time from bootstrap replicates, as in the reference's
    ``test.bs/bs_conf_int.ipynb``: per replicate the arg-max split over the scan, then a
    Student-t interval of those maxima.  ``llk`` is ``[n_split, n_rep]`` (row r = candidate r)."""
    from scipy import stats
    llk = np.asarray(llk, dtype=float)
    best = np.asarray(split_values, dtype=float)[np.argmax(np.where(np.isfinite(llk), llk, -np.inf), axis=0)]
    n = best.size
    mean, sd = best.mean(), best.std(ddof=1) if n > 1 else 0.0
    half = stats.t.ppf(0.5 + level / 2, n - 1) * sd / np.sqrt(n) if n > 1 else 0.0
    return mean, (mean - half, mean + half), best


def _t_interval(b, level):
    from scipy import stats
    m = b.size
    mean, sd = b.mean(), b.std(ddof=1) if m > 1 else 0.0
    half = stats.t.ppf(0.5 + level / 2, m - 1) * sd / np.sqrt(m) if m > 1 else 0.0
    return mean, (mean - half, mean + half), b


def bootstrap_scan_dev(engine, split_values, jsfs_rows, params=None, level=0.95, group=None):
    """A bootstrap scan that keeps the [n_split x n_rep] likelihood table on the device: one
    ``misti_eval_batch_dev`` over the split values x all replicates, then ``misti_argmax_dev`` per
    replicate; only the n_rep winning indices come back.  Returns what ``bootstrap_split_interval``
    returns (mean, interval, per-replicate best split).  Inside a process group the REPLICATES are dealt to the ranks in
    contiguous blocks (BASELINE config 4: 1 000 replicates = 125 per GPU on a node), each rank scans its block on its own GPU and
    one all_gather of the winning indices follows (``dist.bootstrap_sharded``)."""
    rows = np.asarray(jsfs_rows, dtype=float).reshape(-1, 8)
    if _world(group) > 1:
        from . import dist as mdist
        idx = mdist.bootstrap_sharded(lambda a, b: _scan_best(engine, split_values, rows[a:b], params), rows.shape[0], group=group)
    else:
        idx = _scan_best(engine, split_values, rows, params)
    if (idx < 0).any():
        raise ValueError("a replicate has no finite likelihood over the scan")
    return _t_interval(np.asarray(split_values, dtype=float)[idx], level)


def _scan_best(engine, split_values, jsfs_rows, params=None):
    """Per replicate the index of the best split value, reduced on the device (misti_eval_batch_dev + misti_argmax_dev)."""
    import torch
    dev = torch.device("cuda", engine.device)
    split = torch.as_tensor(np.asarray(split_values, dtype=float), device=dev)
    rows = torch.as_tensor(np.asarray(jsfs_rows, dtype=float).reshape(-1, 8), device=dev).contiguous()
    n, R = split.numel(), rows.shape[0]
    par = None
    if engine.n_param:
        par = torch.as_tensor(np.asarray(params, dtype=float).reshape(n, engine.n_param), device=dev).contiguous()
    llk = torch.empty((n, R), dtype=torch.float64, device=dev)
    best = torch.empty(R, dtype=torch.int32, device=dev)
    torch.cuda.current_stream(dev).synchronize()      # the engine's stream is non-blocking: inputs must have landed
    engine.evaluate_dev(n, split.data_ptr(), par.data_ptr() if par is not None else 0, R, rows.data_ptr(), llk.data_ptr())
    engine.argmax_dev(n, R, llk.data_ptr(), best.data_ptr())
    engine.sync()
    return best.cpu().numpy().astype(np.int64)
