"""Python host of the HIP likelihood engine.

Two layers:

``Engine``
    thin, batch-shaped wrapper of the C ABI (``include/misti_hip.h``): one
    model (merged PSMC grid, band/pulse structure, flags) on one GPU, evaluated
    for a batch of candidates x bootstrap JSFS replicates.

``MigrationInference``
    mirror of the reference class of the same name
    (``/root/reference/MigrationInference.py:35-739``): same constructor
    signature, keyword flags, attributes (``.JAFS .lc .lh .mi .pu .Pr .llh
    .times .numT .splitT .dataJAFS .llh_const``), ``JAFSLikelihood(mu)``,
    ``Solve(tol)``, counters and messages - so callers such as ``MiSTI.py`` and
    ``migrationIO.OutputMigration`` work unchanged on top of the GPU path.

There is no CPU path here: if ``libmisti_hip.so`` cannot be loaded or no HIP
device is present, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import math
import sys

import numpy as np

from . import _lib
from ._lib import CPFIT, SMOOTH, TRUE_EPS, UNFOLDED, MistiError, STATUS_TEXT

__all__ = ["Engine", "MigrationInference", "ModelError", "BatchResult"]


class ModelError(SystemExit):
    """Raised where the reference prints an error and calls ``sys.exit(0)``
    (``PrintError``, MigrationInference.py:300-303).  Uncaught, the process exits
    with status 0 exactly as the reference does; a caller may also catch it."""

    def __init__(self, func, text):
        self.message = "MigrationInference class error in function %s(): %s" % (func, text)
        print(self.message)
        super().__init__(0)


class BatchResult:
    """Outputs of one batch: ``llk[C][R]``, ``jafs[C][7]``, ``status[C]`` and, if
    requested, ``lc[C][numT+1][2]`` and ``pr[C][numT+2][6]``."""
    __slots__ = ("llk", "jafs", "status", "lc", "pr", "runaway")

    def __init__(self, llk, jafs, status, lc=None, pr=None, runaway=None):
        self.llk, self.jafs, self.status, self.lc, self.pr = llk, jafs, status, lc, pr
        self.runaway = runaway      # largest corrected rate x interval length (>= ~5: reference-indeterminate)

    @property
    def fraction_failed(self):
        return float((self.status != 0).mean()) if self.status.size else 0.0


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _model_struct(times, lh, bands, pulses, n_param, cpfit, true_eps, smooth, unfolded, sample_date, mixture_th):
    """(misti_model_t, objects that must stay alive while it is used) from the constructor arguments Engine / Lanes share."""
    times = _f64(times)
    lh = _f64(lh)
    numT = int(lh.shape[0])
    if lh.ndim != 2 or lh.shape[1] != 2 or times.shape != (numT - 1,):
        raise ValueError("times must have numT-1 entries and lh shape [numT][2]")
    flags = (CPFIT if cpfit else 0) | (TRUE_EPS if true_eps else 0) | (SMOOTH if smooth else 0) | (UNFOLDED if unfolded else 0)
    bands, pulses = list(bands), list(pulses)
    b_arr = (_lib.Band * max(1, len(bands)))()
    for i, (pop, start, end, value, param) in enumerate(bands):
        b_arr[i] = _lib.Band(int(pop), int(start), int(end), int(param), float(value))
    p_arr = (_lib.Pulse * max(1, len(pulses)))()
    for i, (pop, time, value, param) in enumerate(pulses):
        p_arr[i] = _lib.Pulse(int(pop), int(time), int(param), 0, float(value))
    m = _lib.Model(numT, int(sample_date), flags, len(bands), len(pulses), int(n_param), float(mixture_th),
                   times.ctypes.data_as(C.POINTER(C.c_double)), lh.ctypes.data_as(C.POINTER(C.c_double)), b_arr, p_arr)
    return m, (times, lh, b_arr, p_arr)


class Engine:
    """One model on one GPU (``misti_create`` ... ``misti_destroy``).

    Parameters
    ----------
    times : [numT-1] interval lengths;  lh : [numT][2] PSMC rates.
    bands : iterable of ``(pop, start, end, value, param)`` with pop in {0,1},
        ``end == -1`` meaning "the candidate's split index" and ``param`` the index
        into the candidate parameter vector or -1 for a fixed rate.
    pulses : iterable of ``(pop, time, value, param)``.
    """

    def __init__(self, times, lh, bands=(), pulses=(), n_param=0, cpfit=False, true_eps=False, smooth=False,
                 unfolded=False, sample_date=0, mixture_th=0.0, device=0):
        self._ctx = C.c_void_p()
        self._lib = _lib.load()
        times = _f64(times)
        lh = _f64(lh)
        self.numT = int(lh.shape[0])
        if lh.ndim != 2 or lh.shape[1] != 2 or times.shape != (self.numT - 1,):
            raise ValueError("times must have numT-1 entries and lh shape [numT][2]")
        self.n_param = int(n_param)
        self.flags = (CPFIT if cpfit else 0) | (TRUE_EPS if true_eps else 0) | (SMOOTH if smooth else 0) | (UNFOLDED if unfolded else 0)
        self.unfolded = bool(unfolded)
        bands = list(bands)
        pulses = list(pulses)
        self.n_band = len(bands)
        b_arr = (_lib.Band * max(1, len(bands)))()
        for i, (pop, start, end, value, param) in enumerate(bands):
            b_arr[i] = _lib.Band(int(pop), int(start), int(end), int(param), float(value))
        p_arr = (_lib.Pulse * max(1, len(pulses)))()
        for i, (pop, time, value, param) in enumerate(pulses):
            p_arr[i] = _lib.Pulse(int(pop), int(time), int(param), 0, float(value))
        m = _lib.Model(self.numT, int(sample_date), self.flags, len(bands), len(pulses), self.n_param, float(mixture_th),
                       times.ctypes.data_as(C.POINTER(C.c_double)), lh.ctypes.data_as(C.POINTER(C.c_double)),
                       b_arr, p_arr)
        ctx = C.c_void_p()
        _lib.check(self._lib.misti_create(C.byref(m), int(device), C.byref(ctx)))
        self._ctx = ctx
        self._pid = os.getpid()
        self.device = int(device)

    # -- lifetime --------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            # a context belongs to the process that created it: in a child forked later (a worker pool of the caller)
            # the HIP runtime is not usable - a garbage-collected copy of this object there must not call into it
            if getattr(self, "_pid", None) == os.getpid() and not getattr(self, "_borrowed", False):     # a lane of a Lanes object belongs to it
                self._lib.misti_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- host-buffer evaluation ----------------------------------------------
    def evaluate(self, split_time, params=None, jsfs=None, want_lc=False, want_pr=False, band_bounds=None):
        """``misti_eval_batch``: NumPy in, NumPy out (copies over PCIe).

        ``band_bounds`` ([n][n_band][2] ints, optional): per-candidate (start, end) of every band, replacing the
        model's (``end == -1``: the candidate's split index) - the README's ``::: st ... ::: mc ...`` sweep in one call."""
        split = _f64(np.atleast_1d(split_time))
        n = split.shape[0]
        P = self.n_param
        par = None
        if P:
            par = _f64(params, (n, P))
        rows = _f64(jsfs, (-1, 8)) if jsfs is not None and len(jsfs) else np.zeros((0, 8))
        R = rows.shape[0]
        llk = np.empty((n, R))
        jafs = np.empty((n, 7))
        status = np.empty(n, dtype=np.int32)
        lc = np.empty((n, self.numT + 1, 2)) if want_lc else None
        pr = np.empty((n, self.numT + 2, 6)) if want_pr else None
        ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None
        bb = None
        if band_bounds is not None and self.n_band:
            bb = np.ascontiguousarray(band_bounds, dtype=np.int32).reshape(n, self.n_band, 2)
        _lib.check(self._lib.misti_eval_batch(self._ctx, n, ptr(split), ptr(par), ptr(bb), R, ptr(rows), ptr(llk), ptr(jafs),
                                              ptr(lc), ptr(pr), ptr(status)))
        return BatchResult(llk, jafs, status, lc, pr, self.last_diag(n) if n else np.zeros(0))

    def forward_rates(self, split_time, params=None, want_pr=False, hold_mu=False):
        """``misti_forward_rates``: the model's rates taken as the truth -> the rates a single-genome
        analysis would see under each candidate's migration model (CoalescentRates, :542-564).
        ``hold_mu=True`` reproduces the reference exactly (every interval evaluated with the
        migration rates of the last two-population interval, see include/misti_hip.h).
        Returns (lh[n][numT+1][2], pr[n][numT+2][6] or None, status[n])."""
        split = _f64(np.atleast_1d(split_time))
        n = split.shape[0]
        par = _f64(params, (n, self.n_param)) if self.n_param else None
        lh = np.empty((n, self.numT + 1, 2))
        pr = np.empty((n, self.numT + 2, 6)) if want_pr else None
        status = np.empty(n, dtype=np.int32)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None
        _lib.check(self._lib.misti_forward_rates(self._ctx, n, ptr(split), ptr(par), 1 if hold_mu else 0, ptr(lh), ptr(pr), ptr(status)))
        return lh, pr, status

    def last_diag(self, n_cand):
        """Largest corrected rate x interval length per candidate of the last batch (misti_last_diag)."""
        out = np.empty(int(n_cand))
        _lib.check(self._lib.misti_last_diag(self._ctx, int(n_cand), out.ctypes.data_as(C.c_void_p)))
        return out

    # -- device-buffer evaluation (torch tensors on this device) ------------------
    def use_stream(self, stream_handle):
        """Issue kernels on an existing hipStream_t (int handle), e.g.
        ``torch.cuda.current_stream().cuda_stream``; ``None`` restores the own stream."""
        _lib.check(self._lib.misti_set_stream(self._ctx, C.c_void_p(stream_handle) if stream_handle else None))

    def stream_handle(self):
        """The hipStream_t (int) this engine issues on - its own non-blocking stream unless replaced."""
        h = C.c_void_p()
        _lib.check(self._lib.misti_get_stream(self._ctx, C.byref(h)))
        return h.value or 0

    def set_hints(self, integer_splits=False):
        """``misti_set_hints``: what the caller knows about the batches it will issue on this context (device-buffer form: the split times live in
        HBM).  ``integer_splits``: no split time has a fractional part - the launch that only serves fractional splits is not made (a batch that
        has one after all gets status 4 for that candidate, never a wrong value)."""
        _lib.check(self._lib.misti_set_hints(self._ctx, _lib.HINT_INTEGER_SPLITS if integer_splits else 0))

    def evaluate_dev(self, n_cand, d_split, d_params, n_rep, d_jsfs, d_llk, d_jafs=0, d_lc=0, d_pr=0, d_status=0, d_bounds=0):
        """``misti_eval_batch_dev``: raw device addresses (ints); asynchronous."""
        v = lambda p: C.c_void_p(int(p)) if p else None
        _lib.check(self._lib.misti_eval_batch_dev(self._ctx, int(n_cand), v(d_split), v(d_params), v(d_bounds), int(n_rep), v(d_jsfs),
                                                  v(d_llk), v(d_jafs), v(d_lc), v(d_pr), v(d_status)))

    def llk_dev(self, n_cand, d_jafs, d_status, n_rep, d_jsfs, d_llk):
        v = lambda p: C.c_void_p(int(p)) if p else None
        _lib.check(self._lib.misti_llk_dev(self._ctx, int(n_cand), v(d_jafs), v(d_status), int(n_rep), v(d_jsfs), v(d_llk)))

    def argmax_dev(self, n_cand, n_rep, d_llk, d_best, d_best_llk=0):
        """``misti_argmax_dev``: per replicate the index of the best candidate (raw device addresses; asynchronous)."""
        v = lambda p: C.c_void_p(int(p)) if p else None
        _lib.check(self._lib.misti_argmax_dev(self._ctx, int(n_cand), int(n_rep), v(d_llk), v(d_best), v(d_best_llk)))

    def sync(self):
        _lib.check(self._lib.misti_sync(self._ctx))

    def nm_solve(self, starts, split_time, jsfs_row, tol=1e-4, maxiter=1000):
        """``misti_nm_solve``: SciPy-exact Nelder-Mead from every row of ``starts`` with the simplices resident in
        HBM (``MigrationInference.Solve`` for many starts; reference semantics :718-733).
        Returns dict(x[S][P], llh[S], nit[S], nfev[S], status[S]) - status 0 converged, 2 iteration budget."""
        st = _f64(starts, (-1, self.n_param))
        S = st.shape[0]
        row = _f64(jsfs_row, (8,))
        x = np.empty((S, self.n_param))
        llh = np.empty(S)
        nit, nfev, status = (np.empty(S, dtype=np.int32) for _ in range(3))
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.misti_nm_solve(self._ctx, S, ptr(st), float(split_time), ptr(row), float(tol), float(tol), int(maxiter),
                                            ptr(x), ptr(llh), ptr(nit), ptr(nfev), ptr(status)))
        stats = (C.c_int64 * 2)()
        _lib.check(self._lib.misti_nm_last_stats(self._ctx, stats))
        spec = C.c_int64(0)
        _lib.check(self._lib.misti_nm_last_spec_iterations(self._ctx, C.byref(spec)))
        return dict(x=x, llh=llh, nit=nit, nfev=nfev, status=status, iterations_issued=int(stats[0]), slots=int(stats[1]),
                    speculative_iterations=int(spec.value))

    def basinhopping(self, starts, split_time, jsfs_row, rngs, niter=100, T=0.5, stepsize=0.5, interval=50, target_accept_rate=0.5,
                     stepwise_factor=0.9, xatol=1e-4, fatol=1e-4, nm_maxiter=None, nm_maxfev=None):
        """``misti_basinhopping``: ``scipy.optimize.basinhopping(-JAFSLikelihood, x0, niter, T, stepsize,
        minimizer_kwargs=dict(method='Nelder-Mead'), rng=rngs[s])`` from every row of ``starts`` at once - the reference's
        ``Solve(globalOpt=True)`` (``/root/reference/MigrationInference.py:723-725``, T = 0.5) for many starts.
        ``rngs``: one ``numpy.random.Generator`` per start (or one seed per start): the uniforms SciPy would draw from it
        (per hop ``n_param`` for the displacement, one for the Metropolis test) are drawn here, up front.
        Returns dict(x[S][P], llh[S], nfev[S], failures[S], accepted[S])."""
        st = _f64(starts, (-1, self.n_param))
        S, N = st.shape
        row = _f64(jsfs_row, (8,))
        uni = draw_uniforms(rngs, S, int(niter), N)
        x = np.empty((S, N))
        llh = np.empty(S)
        nfev, failures, accepted = (np.empty(S, dtype=np.int32) for _ in range(3))
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.misti_basinhopping(self._ctx, S, ptr(st), float(split_time), ptr(row), int(niter), float(T), float(stepsize), int(interval),
                                                float(target_accept_rate), float(stepwise_factor), float(xatol), float(fatol),
                                                int(nm_maxiter if nm_maxiter is not None else 200 * N), int(nm_maxfev if nm_maxfev is not None else 200 * N),
                                                ptr(uni), ptr(x), ptr(llh), ptr(nfev), ptr(failures), ptr(accepted)))
        stats = (C.c_int64 * 2)()
        _lib.check(self._lib.misti_nm_last_stats(self._ctx, stats))
        return dict(x=x, llh=llh, nfev=nfev, failures=failures, accepted=accepted, iterations_issued=int(stats[0]), slots=int(stats[1]))

    def enable_solver_trace(self, on=True):
        """Record, for the following batches, SciPy-comparable solver statistics per candidate and interval
        (``misti_enable_solver_trace``; see ``solver_trace``)."""
        _lib.check(self._lib.misti_enable_solver_trace(self._ctx, 1 if on else 0))

    def solver_trace(self, n_cand, cand=None):
        """Solver trace of the last batch: dict of ``nfev``, ``status``, ``kind`` arrays ``[n_cand][numT+1]``
        (kind 0 none, 1 closed form, 2 bounded TRF, 3 unbounded TRF; status = SciPy's termination code; noise 1 where a
        default-fit solve went on past a gradient test only the reference's noisy residual would have failed; stall 1 where the stall rule returned the
        starting point: status 3 and nfev 1, the reference needs 14 - 23 evaluations to get nowhere) and, with
        ``cand`` given (batches of at most 64 candidates), ``iterates [numT][200][2]``: the trial points of the
        unbounded solves of that candidate's chain (NaN beyond nfev)."""
        n = int(n_cand)
        words = np.empty((n, self.numT + 1), dtype=np.int32)
        its = np.empty((self.numT, _lib.TRACE_MAX_ITER, 2)) if cand is not None else None
        _lib.check(self._lib.misti_last_solver_trace(self._ctx, n, words.ctypes.data_as(C.c_void_p), int(cand) if cand is not None else 0,
                                                     its.ctypes.data_as(C.c_void_p) if its is not None else None))
        out = {"nfev": words & 0xffff, "status": (words >> 16) & 15, "kind": (words >> 20) & 15, "noise": (words >> 24) & 1, "stall": (words >> 25) & 1}
        if its is not None:
            out["iterates"] = its
        return out

    def enable_timing(self, on=True):
        _lib.check(self._lib.misti_enable_timing(self._ctx, 1 if on else 0))

    def kernel_times(self, reset=False):
        """(ms per kernel, launches per kernel) accumulated from HIP events around every launch."""
        ms = (C.c_double * 3)()
        n = (C.c_int64 * 3)()
        _lib.check(self._lib.misti_kernel_times(self._ctx, ms, n, 1 if reset else 0))
        names = ("correct", "spectrum", "llk")
        return {k: ms[i] for i, k in enumerate(names)}, {k: n[i] for i, k in enumerate(names)}


# ------------------------------------------------------------------------------
class Lanes:
    """``misti_create_lanes`` ... ``misti_destroy_lanes``: n engine contexts of one model on one device, each with its own non-blocking
    stream, so that independent batches overlap on the GPU (include/misti_hip.h, "lanes"; the overlapped rate of the headline benchmark).
    The pool itself lives in the library since round 6 - ``misti_amd.lanes.LanePool`` and ``bench.py`` sit on this class.  Same constructor
    as ``Engine`` plus ``lanes``.  Device-buffer form only (raw device addresses, asynchronous); results are bit for bit a single
    context's."""

    def __init__(self, times, lh, bands=(), pulses=(), n_param=0, cpfit=False, true_eps=False, smooth=False,
                 unfolded=False, sample_date=0, mixture_th=0.0, device=0, lanes=20):
        self._h = C.c_void_p()
        self._lib = _lib.load()
        m, keep = _model_struct(times, lh, bands, pulses, n_param, cpfit, true_eps, smooth, unfolded, sample_date, mixture_th)
        self.numT, self.n_param, self.n_band, self.device = int(m.numT), int(n_param), int(m.n_band), int(device)
        h = C.c_void_p()
        _lib.check(self._lib.misti_create_lanes(C.byref(m), int(device), int(lanes), C.byref(h)))
        self._h = h
        self._pid = os.getpid()
        self.n_lanes = int(self._lib.misti_lanes_size(h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if getattr(self, "_pid", None) == os.getpid():
                self._lib.misti_destroy_lanes(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def engine(self, i):
        """Lane i's context as an ``Engine`` that does NOT own it (timing, stream handle, solver trace, ``last_diag`` per lane)."""
        ctx = C.c_void_p()
        _lib.check(self._lib.misti_lanes_context(self._h, int(i), C.byref(ctx)))
        e = Engine.__new__(Engine)
        e._lib, e._ctx, e._pid, e._borrowed = self._lib, ctx, self._pid, True
        e.numT, e.n_param, e.n_band, e.device, e.unfolded = self.numT, self.n_param, self.n_band, self.device, False
        return e

    def set_hints(self, integer_splits=False):
        _lib.check(self._lib.misti_lanes_set_hints(self._h, _lib.HINT_INTEGER_SPLITS if integer_splits else 0))

    def evaluate_dev(self, lane, n_cand, d_split, d_params, n_rep, d_jsfs, d_llk, d_jafs=0, d_lc=0, d_pr=0, d_status=0, d_bounds=0):
        """``misti_lanes_eval_batch_dev``: one batch on lane ``lane`` (``None`` / -1: an idle lane, else round-robin); returns the lane used."""
        v = lambda p: C.c_void_p(int(p)) if p else None
        used = C.c_int(-1)
        _lib.check(self._lib.misti_lanes_eval_batch_dev(self._h, _lib.LANE_ANY if lane is None else int(lane), int(n_cand), v(d_split), v(d_params), v(d_bounds),
                                                        int(n_rep), v(d_jsfs), v(d_llk), v(d_jafs), v(d_lc), v(d_pr), v(d_status), C.byref(used)))
        return int(used.value)

    def bind_dev(self, lane, n_cand, d_split, d_params, n_rep, d_jsfs, d_llk, d_jafs=0, d_lc=0, d_pr=0, d_status=0, d_bounds=0):
        """``evaluate_dev`` with its arguments converted ONCE: returns a function of no arguments that issues this batch on lane ``lane``
        (an explicit lane).  A sweep that re-issues a batch on fixed device buffers - the bench's steps - saves the ~2 us of ctypes argument
        conversion per call (12.6 -> 10.8 us per step on the headline grid; the rest is the library's three launches and two event records)."""
        v = lambda p: C.c_void_p(int(p)) if p else None
        fn, check = self._lib.misti_lanes_eval_batch_dev, _lib.check
        args = (self._h, C.c_int(int(lane)), C.c_int64(int(n_cand)), v(d_split), v(d_params), v(d_bounds), C.c_int64(int(n_rep)), v(d_jsfs),
                v(d_llk), v(d_jafs), v(d_lc), v(d_pr), v(d_status), None)

        def issue():
            r = fn(*args)
            if r:
                check(r)
        return issue

    def wait(self, lane):
        _lib.check(self._lib.misti_lanes_wait(self._h, int(lane)))

    def busy(self, lane):
        r = self._lib.misti_lanes_busy(self._h, int(lane))
        if r < 0:
            _lib.check(r)
        return bool(r)

    def sync(self):
        _lib.check(self._lib.misti_lanes_sync(self._h))


# ------------------------------------------------------------------------------
class MultiEngine:
    """One model on a LIST of devices from one process (``misti_create_multi`` ... ``misti_destroy_multi``): what a node's 1, 2, 4
    or 8 GPUs are to a caller of the C ABI - the reference's counterpart is ``parallel -j 20 ./MiSTI.py ...``
    (``/root/reference/README.md:110-115``).  Same constructor as ``Engine`` with ``devices`` (a list; an entry may repeat) in
    place of ``device``.  ``evaluate`` deals whole lambda-correction chains to the devices and returns exactly what ``Engine.evaluate``
    returns; ``nm_solve`` / ``basinhopping`` deal contiguous blocks of starts.  Nothing here falls back to fewer devices: a
    device that cannot be opened fails the constructor."""

    def __init__(self, times, lh, bands=(), pulses=(), n_param=0, cpfit=False, true_eps=False, smooth=False,
                 unfolded=False, sample_date=0, mixture_th=0.0, devices=(0,)):
        self._m = C.c_void_p()
        self._lib = _lib.load()
        times = _f64(times)
        lh = _f64(lh)
        self.numT = int(lh.shape[0])
        if lh.ndim != 2 or lh.shape[1] != 2 or times.shape != (self.numT - 1,):
            raise ValueError("times must have numT-1 entries and lh shape [numT][2]")
        self.n_param = int(n_param)
        self.flags = (CPFIT if cpfit else 0) | (TRUE_EPS if true_eps else 0) | (SMOOTH if smooth else 0) | (UNFOLDED if unfolded else 0)
        self.unfolded = bool(unfolded)
        bands, pulses = list(bands), list(pulses)
        self.n_band = len(bands)
        b_arr = (_lib.Band * max(1, len(bands)))()
        for i, (pop, start, end, value, param) in enumerate(bands):
            b_arr[i] = _lib.Band(int(pop), int(start), int(end), int(param), float(value))
        p_arr = (_lib.Pulse * max(1, len(pulses)))()
        for i, (pop, time, value, param) in enumerate(pulses):
            p_arr[i] = _lib.Pulse(int(pop), int(time), int(param), 0, float(value))
        m = _lib.Model(self.numT, int(sample_date), self.flags, len(bands), len(pulses), self.n_param, float(mixture_th),
                       times.ctypes.data_as(C.POINTER(C.c_double)), lh.ctypes.data_as(C.POINTER(C.c_double)), b_arr, p_arr)
        self.devices = [int(d) for d in devices]
        dev = (C.c_int32 * max(1, len(self.devices)))(*self.devices)
        h = C.c_void_p()
        _lib.check(self._lib.misti_create_multi(C.byref(m), len(self.devices), dev, C.byref(h)))
        self._m = h
        self._pid = os.getpid()

    def close(self):
        if getattr(self, "_m", None) is not None and self._m:
            if getattr(self, "_pid", None) == os.getpid():
                self._lib.misti_destroy_multi(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def evaluate(self, split_time, params=None, jsfs=None, want_lc=False, want_pr=False, band_bounds=None):
        """``misti_multi_eval_batch``: as ``Engine.evaluate`` (``runaway`` is not collected across devices: None)."""
        split = _f64(np.atleast_1d(split_time))
        n = split.shape[0]
        P = self.n_param
        par = _f64(params, (n, P)) if P else None
        rows = _f64(jsfs, (-1, 8)) if jsfs is not None and len(jsfs) else np.zeros((0, 8))
        R = rows.shape[0]
        llk = np.empty((n, R))
        jafs = np.empty((n, 7))
        status = np.empty(n, dtype=np.int32)
        lc = np.empty((n, self.numT + 1, 2)) if want_lc else None
        pr = np.empty((n, self.numT + 2, 6)) if want_pr else None
        ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None
        bb = None
        if band_bounds is not None and self.n_band:
            bb = np.ascontiguousarray(band_bounds, dtype=np.int32).reshape(n, self.n_band, 2)
        _lib.check(self._lib.misti_multi_eval_batch(self._m, n, ptr(split), ptr(par), ptr(bb), R, ptr(rows), ptr(llk), ptr(jafs),
                                                    ptr(lc), ptr(pr), ptr(status)))
        return BatchResult(llk, jafs, status, lc, pr, None)

    def last_shards(self):
        """(candidates per context, chains per context) of the last ``evaluate``."""
        D = len(self.devices)
        a, b = (C.c_int64 * D)(), (C.c_int64 * D)()
        _lib.check(self._lib.misti_multi_last_shards(self._m, a, b))
        return list(a), list(b)

    def last_cost(self):
        """Summed chain cost per context of the last ``evaluate`` (what the dealing balances: chain length + members / 64)."""
        c = (C.c_double * len(self.devices))()
        _lib.check(self._lib.misti_multi_last_cost(self._m, c))
        return list(c)

    def evaluate_dev_gathered(self, n_cand, rows_per_shard, d_split, d_params, n_rep, d_jsfs, d_llk_all, d_status_all=None, d_bounds=None):
        """``misti_multi_eval_batch_dev``: context i evaluates ``n_cand[i]`` candidates from device pointers on ITS device and the
        log-likelihoods are all-gathered on the devices by RCCL inside the library (every ``d_llk_all[i]`` ends as the whole
        ``[D][rows_per_shard][n_rep]`` table).  Arguments are lists (one entry per context) of integer device addresses, e.g.
        ``tensor.data_ptr()``.  Asynchronous; ``sync()`` waits for every context's stream."""
        D = len(self.devices)
        arr = lambda ptrs: (C.c_void_p * D)(*[C.c_void_p(int(p)) if p else None for p in ptrs])
        n = (C.c_int64 * D)(*[int(v) for v in n_cand])
        _lib.check(self._lib.misti_multi_eval_batch_dev(self._m, n, int(rows_per_shard), arr(d_split), arr(d_params) if d_params is not None else None,
                                                        arr(d_bounds) if d_bounds is not None else None, int(n_rep), arr(d_jsfs), arr(d_llk_all),
                                                        arr(d_status_all) if d_status_all is not None else None))

    def sync(self):
        _lib.check(self._lib.misti_multi_sync(self._m))

    def nm_solve(self, starts, split_time, jsfs_row, tol=1e-4, maxiter=1000):
        """``misti_multi_nm_solve``: ``Engine.nm_solve`` with the starts dealt to the devices in contiguous blocks."""
        st = _f64(starts, (-1, self.n_param))
        S = st.shape[0]
        row = _f64(jsfs_row, (8,))
        x = np.empty((S, self.n_param))
        llh = np.empty(S)
        nit, nfev, status = (np.empty(S, dtype=np.int32) for _ in range(3))
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.misti_multi_nm_solve(self._m, S, ptr(st), float(split_time), ptr(row), float(tol), float(tol), int(maxiter),
                                                  ptr(x), ptr(llh), ptr(nit), ptr(nfev), ptr(status)))
        return dict(x=x, llh=llh, nit=nit, nfev=nfev, status=status)

    def basinhopping(self, starts, split_time, jsfs_row, rngs, niter=100, T=0.5, stepsize=0.5, interval=50, target_accept_rate=0.5,
                     stepwise_factor=0.9, xatol=1e-4, fatol=1e-4, nm_maxiter=None, nm_maxfev=None):
        """``misti_multi_basinhopping``: ``Engine.basinhopping`` with the starts dealt to the devices in contiguous blocks."""
        st = _f64(starts, (-1, self.n_param))
        S, N = st.shape
        row = _f64(jsfs_row, (8,))
        uni = draw_uniforms(rngs, S, int(niter), N)
        x = np.empty((S, N))
        llh = np.empty(S)
        nfev, failures, accepted = (np.empty(S, dtype=np.int32) for _ in range(3))
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.misti_multi_basinhopping(self._m, S, ptr(st), float(split_time), ptr(row), int(niter), float(T), float(stepsize), int(interval),
                                                      float(target_accept_rate), float(stepwise_factor), float(xatol), float(fatol),
                                                      int(nm_maxiter if nm_maxiter is not None else 200 * N), int(nm_maxfev if nm_maxfev is not None else 200 * N),
                                                      ptr(uni), ptr(x), ptr(llh), ptr(nfev), ptr(failures), ptr(accepted)))
        return dict(x=x, llh=llh, nfev=nfev, failures=failures, accepted=accepted)


def draw_uniforms(rngs, S, niter, N):
    """The uniforms SciPy's basin hopping would draw from start s's generator, in its order: per hop N for the displacement
    (RandomDisplacement: rng.uniform(-stepsize, stepsize, shape)), then one for the Metropolis test.  ``rngs``: one
    ``numpy.random.Generator`` or seed per start."""
    gens = [g if isinstance(g, np.random.Generator) else np.random.default_rng(g) for g in rngs]
    if len(gens) != S:
        raise ValueError("one generator (or seed) per start")
    uni = np.empty((S, niter, N + 1))
    for s_, g in enumerate(gens):
        for h in range(niter):
            uni[s_, h, :N] = g.random(N)
            uni[s_, h, N] = g.random()
    return uni


def _classes(v, unfolded):
    """The spectrum classes the likelihood distinguishes: all 7, or folded pairs 0+6, 1+5, 2+4 and 3 (:217-227, :600-609)."""
    return list(v) if unfolded else [v[0] + v[6], v[1] + v[5], v[2] + v[4], v[3]]


def _multinomial_const(counts, unfolded):
    """log of the multinomial coefficient of the data (llh_const of SetJAFS)."""
    c = math.lgamma(sum(counts) + 1)
    for v in _classes(counts, unfolded):          # subtracted one by one, as the reference does (same rounding)
        c -= math.lgamma(v + 1)
    return c


def _multinomial_logterm(counts, spectrum, unfolded):
    return sum(c * math.log(p) for c, p in zip(_classes(counts, unfolded), _classes(spectrum, unfolded)))


def _band_record(el, sample_date, error):
    """One ``-mi pop start end value opt`` option -> (pop 0/1, start, end, value, optimised); SetModel's checks :236-247."""
    pop, start, end, value, opt = int(el[0]) - 1, int(el[1]), int(el[2]), float(el[3]), int(el[4]) == 1
    if pop not in (0, 1):
        error("SetModel", "Population index should be 1 or 2.")
    if start < sample_date:
        error("SetModel", "Migration start (" + str(start) + ") should be larger than or equal to sample date (" + str(sample_date) + ").")
    if end <= start:
        error("SetModel", "Migration start (" + str(start) + ") should be strictly less than migration end (" + str(end) + ").")
    return pop, start, end, value, opt


def _pulse_record(el, sample_date, error):
    """One ``-pu pop time value opt`` option -> (pop 0/1, time, value, optimised); SetModel's checks :259-271."""
    pop, t, value, opt = int(el[0]) - 1, int(el[1]), float(el[2]), int(el[3]) == 1
    if pop not in (0, 1):
        error("SetModel", "Population index should be 1 or 2.")
    if t < sample_date:
        error("SetModel", "Pulse migration time (" + str(t) + ") should be larger than or equal to sample date (" + str(sample_date) + ").")
    if value < 0 or value > 1:
        error("SetModel", "Pulse migration rate should be between 0 and 1.")
    return pop, t, value, opt


class MigrationInference:
    """GPU-backed mirror of the reference class (MigrationInference.py:35-739)."""
    COUNT_LLH = 0
    CORRECTION_CALLED = 0
    CORRECTION_FAILED = 0

    def __init__(self, times, lambdas, dataJAFS, splitT, mi=[], pu=[], **kwargs):
        kw = kwargs
        self.debug = bool(kw.get("debug", False))
        self.enableOutput = self.debug or bool(kw.get("enableOutput", False))
        if self.enableOutput:
            print("MigrationInference: output enabled.")
        self.cpfit = bool(kw.get("cpfit", False))
        self.correct = not bool(kw.get("trueEPS", False))
        self.smooth = bool(kw.get("smooth", False))
        self.LLHpsmc = "Tpsmc" in kw
        if self.LLHpsmc:
            self.Tpsmc = kw["Tpsmc"]
        self.unfolded = bool(kw.get("unfolded", False))
        self.thrh = [1.0, 1.0]
        if "thrh" in kw and len(kw["thrh"]) == 2:
            self.thrh = kw["thrh"]
        self.sampleDate = kw.get("sampleDate", 0)
        self.mixtureTH = kw.get("mixtureTH", 0.0)
        self._device = int(kw.get("device", 0))
        if splitT < self.sampleDate:                                            # :85-86
            self.PrintError("__init__", "cannot initialise class with split time being more recent than sample date.")
        self._split_in = float(splitT)
        self._times0 = [float(v) for v in times]
        self._lh0 = [[float(l[0]), float(l[1])] for l in lambdas]
        frac = splitT % 1                                                       # :89-99, incl. the mutation
        splitT = int(splitT)
        if splitT - 1 > len(times):
            self.PrintError("__init__", "Invalid value for split time, cannot create Migration class instance.")
        if frac != 0.0:
            t1 = frac * times[splitT]
            t2 = times[splitT] - t1
            times[splitT] = t1
            times.insert(splitT + 1, t2)
            lambdas.insert(splitT + 1, lambdas[splitT])
            splitT += 1
        self.lh = list(lambdas)
        self.times = times
        self.numT = len(self.lh)
        if len(self.times) != self.numT - 1:                                    # :105-107
            print("Unexpected number of time intervals")
            raise SystemExit(0)
        self.discr = 1
        self.splitT = splitT
        self.SetModel(mi, pu)
        self.SetJAFS(dataJAFS)
        self.JAFSsize = self.snps
        self.lc = [[1, 1] for _ in range(self.numT)]
        self.JAFS = None
        self.Pr = None
        self.llh = None
        self.status = 0
        self.doPlot = False
        # bands/pulses in C-ABI form; every band is expressed on this candidate's grid
        bands = [(pop, start, end, val, -1) for pop, start, end, val in self._fixedMis]
        bands += [(pop, start, end, val, i) for i, (pop, start, end, val) in enumerate(self.optMis)]
        n_opt_mi = len(self.optMis)
        pulses = [(pop, t, val, -1) for pop, t, val in self._fixedPus]
        pulses += [(pop, t, val, n_opt_mi + i) for i, (pop, t, val) in enumerate(self.optPus)]
        self._engine = Engine(self._times0, self._lh0, bands, pulses, n_param=n_opt_mi + len(self.optPus),
                              cpfit=self.cpfit, true_eps=not self.correct, smooth=self.smooth, unfolded=self.unfolded,
                              sample_date=int(self.sampleDate), mixture_th=float(self.mixtureTH), device=self._device)
        if self.debug:
            print("MigrationInference class initialized. Class size", self.numT)

    # -- reference helpers ---------------------------------------------------------
    def PrintError(self, func, text):                                           # :300-303
        raise ModelError(func, text)

    def SetJAFS(self, dataJAFS, normalize=False):                               # :202-227
        """Data spectrum of 8 numbers (total + 7 classes) and the data-only term of the multinomial likelihood."""
        if len(dataJAFS) != 8:
            self.PrintError("SetJAFS", "Unexpected data SFS.")
        self._row = [float(v) for v in dataJAFS]
        self.dataJAFS = list(dataJAFS[1:])
        self.snps = sum(self.dataJAFS)
        self.llh_const = _multinomial_const(self.dataJAFS, self.unfolded)

    def SetModel(self, mis, pus):                                               # :229-289
        """``-mi`` / ``-pu`` descriptors -> band and pulse records (``_bands``, ``_pulses``; what the C ABI takes)
        and, derived from them, the per-interval tables ``mi`` / ``pu`` and the lists ``optMis`` / ``optPus`` the
        reference's callers read.  Errors are the reference's (message and exit status 0)."""
        self._bands = [_band_record(el, self.sampleDate, self.PrintError) for el in mis]
        self._pulses = [_pulse_record(el, self.sampleDate, self.PrintError) for el in pus]
        for pop, start, end, _, _ in self._bands:
            if end > self.numT:                      # the reference runs off the end of its table here (IndexError)
                self.PrintError("SetModel", "Migration end (" + str(end) + ") is beyond the last time interval (" + str(self.numT) + ").")
        for pop, t, _, _ in self._pulses:
            if t >= self.numT:
                self.PrintError("SetModel", "Pulse migration time (" + str(t) + ") is beyond the last time interval (" + str(self.numT - 1) + ").")
        taken = set()
        for pop, start, end, _, _ in self._bands:
            cells = {(pop, t) for t in range(start, end)}
            if cells & taken:
                self.PrintError("SetModel", "Migration rate intervals should not overlap.")
            taken |= cells
        if len({t for _, t, _, _ in self._pulses}) != len(self._pulses):
            self.PrintError("SetModel", "Current version allows only single-direction pulse migration at a time.")
        self.optMis = [[pop, start, end, val] for pop, start, end, val, opt in self._bands if opt]
        self.optPus = [[pop, t, val] for pop, t, val, opt in self._pulses if opt]
        self._fixedMis = [[pop, start, end, val] for pop, start, end, val, opt in self._bands if not opt]
        self._fixedPus = [[pop, t, val] for pop, t, val, opt in self._pulses if not opt]
        self.optMisSize, self.optPusSize = len(self.optMis), len(self.optPus)
        self._fill_tables([b[3] for b in self.optMis] + [q[2] for q in self.optPus])

    def _fill_tables(self, values):
        """Per-interval tables from the records; optimised bands / pulses take ``values`` (bands first, in option order)."""
        self.mi = [[0.0, 0.0] for _ in range(self.numT)]
        self.pu = [[0.0, 0.0] for _ in range(self.numT)]
        free = iter(values)
        rate_of = {}
        for k, (pop, start, end, val, opt) in enumerate(self._bands):
            if opt:
                rate_of[k] = next(free)
        for k, (pop, start, end, val, opt) in enumerate(self._bands):
            for t in range(start, end):
                self.mi[t][pop] = rate_of.get(k, val)
        for pop, t, val, opt in self._pulses:
            self.pu[t][pop] = next(free) if opt else val

    def MapParameters(self, params):                                            # :291-298
        if len(params) != self.optMisSize + self.optPusSize:
            self.PrintError("MapParameters", "Incorrect number of parameters.")
        self._fill_tables(list(params))

    # -- the hot path ----------------------------------------------------------------
    def JAFSLikelihood(self, mu):                                               # :566-614
        cls = MigrationInference
        cls.COUNT_LLH += 1
        self.llh = -10 ** 9
        for v in mu:
            if v < 0:
                print("Hit negative value of migration rate")
                self.status = 1
                return -np.inf
        self.MapParameters(mu)
        self._mu = [float(v) for v in mu]
        cls.CORRECTION_CALLED += 1
        res = self._engine.evaluate([self._split_in], [list(mu)] if len(mu) else None, [self._row], want_lc=True, want_pr=True)
        self.status = int(res.status[0])
        if self.status == 2:
            cls.CORRECTION_FAILED += 1
            print("Lambda correction failed")
            return -np.inf
        if self.status == 3:
            self.PrintError("JAFSpectrum", "Infinite coalescent time. No migration.")
        if self.status != 0:
            # non-finite intermediate / stiff interval: the candidate has no value here
            print("MigrationInference: " + STATUS_TEXT.get(self.status, "status %d" % self.status))
            return -np.inf
        self.runaway = float(res.runaway[0])
        if self.runaway >= 5.0 and self.enableOutput:
            print("MigrationInference: corrected rate x interval length reaches %.3g - the lambda-correction ran into its "
                  "flat regime; the reference's value for such a candidate is not determined to 1e-9" % self.runaway)
        self.lc = [[float(a), float(b)] for a, b in res.lc[0][: self.numT]]
        self.Pr = [[[float(r[0]), float(r[1])], [float(r[2]), float(r[3])], [float(r[4]), float(r[5])]]
                   for r in res.pr[0][: self.splitT + 1]]
        self.JAFS = [float(v) for v in res.jafs[0]]
        self.llh = float(res.llk[0, 0])
        return self.llh

    def CoalescentRates(self):                                                  # :542-564
        """Forward map (TestModel route): ``lc`` := the current rates, ``lh`` := what PSMC would see
        under the current migration model, ``Pr`` := the pair-state trace.  As in the reference, every
        interval is evaluated with the migration rates of the last two-population interval (its
        CorrectLambda object keeps them from the preceding likelihood call); for a per-interval
        forward map use ``Engine.forward_rates``."""
        mu = getattr(self, "_mu", None)
        if mu is None:
            mu = [b[3] for b in self.optMis] + [q[2] for q in self.optPus]
        lh, pr, status = self._engine.forward_rates([self._split_in], [list(mu)] if len(mu) else None, want_pr=True, hold_mu=True)
        for i in range(self.numT):
            self.lc[i] = [self.lh[i][0], self.lh[i][1]]
        for t in range(min(self.splitT, self.numT - 1)):
            self.lh[t] = [float(lh[0][t][0]), float(lh[0][t][1])]
        self.Pr = [[[float(r[0]), float(r[1])], [float(r[2]), float(r[3])], [float(r[4]), float(r[5])]]
                   for r in pr[0][: self.splitT + 1]]

    def JAFSLikelihoodBatch(self, split_times, params=None, jsfs_rows=None, **kw):
        """Batch form without a reference counterpart: many candidates x replicates in
        one call on the same grid and band structure (bands ending at this object's
        split follow each candidate's own split)."""
        return self._engine.evaluate(split_times, params, [self._row] if jsfs_rows is None else jsfs_rows, **kw)

    def MaximumLLHFunction(self):                                               # :696-711
        """Likelihood of the data under its own (saturated) spectrum: the upper bound of JAFSLikelihood."""
        tot = sum(self.dataJAFS)
        return self.llh_const + _multinomial_logterm(self.dataJAFS, [v / tot for v in self.dataJAFS], self.unfolded)

    def ObjectiveFunction(self, mu):                                            # :713-716
        res = -self.JAFSLikelihood(mu)
        print(mu, res)
        return res

    def Solve(self, tol=1e-4, globalOpt=False):                                 # :718-733
        if self.optMisSize + self.optPusSize > 0:
            from scipy import optimize
            init = [v[3] for v in self.optMis] + [v[2] for v in self.optPus]
            if globalOpt:
                res = optimize.basinhopping(self.ObjectiveFunction, init, T=0.5, minimizer_kwargs=dict(method="Nelder-Mead"))
            else:
                res = optimize.minimize(self.ObjectiveFunction, init, method="Nelder-Mead",
                                        options={"xatol": tol, "fatol": tol, "maxiter": 1000, "disp": True})
            return [res.x, -res.fun]
        return [[], self.JAFSLikelihood([])]

    @staticmethod
    def Report():                                                               # :735-739
        print("Total number of likelihood function calls is", MigrationInference.COUNT_LLH)
        print("Lambda correction called", MigrationInference.CORRECTION_CALLED, "times.")
        print("Lambda correction failed", MigrationInference.CORRECTION_FAILED, "times.")


def truth_spectrum(times, lh, split, bands, pulses, sample_date, device=0):
    """Expected JSFS of a fully specified model (``--trueEPS`` route of TestModel.py:95-96)
    computed on the GPU; used to synthesise data spectra for the workloads."""
    with Engine(times, lh, bands, pulses, 0, cpfit=True, true_eps=True, smooth=False, unfolded=True,
                sample_date=sample_date, device=device) as e:
        r = e.evaluate([float(split)])
    if int(r.status[0]) != 0:
        raise MistiError(int(r.status[0]), "truth spectrum failed: " + STATUS_TEXT.get(int(r.status[0]), "?"))
    return [float(v) for v in r.jafs[0]]
