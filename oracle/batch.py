"""Batch helpers around the CPU oracle (TEST INFRASTRUCTURE: used by tests/, smoke()
and the cpu_baseline leg of bench.py only)."""
from __future__ import annotations

import time
import warnings

import numpy as np

from .misti_oracle import OracleModel, llh_const


def _mis_pus(bands, pulses, split, params):
    mis, pus = [], []
    for pop, start, end, value, param in bands:
        v = params[param] if param >= 0 else value
        mis.append([pop + 1, start, split if end < 0 else end, v, 1 if param >= 0 else 0])
    for pop, t, value, param in pulses:
        v = params[param] if param >= 0 else value
        pus.append([pop + 1, t, v, 1 if param >= 0 else 0])
    return mis, pus


def oracle_eval(times, lh, bands, pulses, flags, sample_date, split, params, jsfs_rows, mixture_th=0.0):
    """One candidate x all replicates with the oracle: (llk[R], jafs[7] or None, status)."""
    params = [] if params is None else [float(v) for v in params]
    import math
    s_int = int(math.floor(split))
    split_idx = s_int + (1 if split != s_int else 0)
    mis, pus = _mis_pus(bands, pulses, split_idx, params)
    order = [p for b in bands for p in [b[4]] if p >= 0] + [p for b in pulses for p in [b[3]] if p >= 0]
    mu = [params[i] for i in order]
    rows = np.atleast_2d(np.asarray(jsfs_rows, dtype=float)) if jsfs_rows is not None and len(jsfs_rows) else np.zeros((0, 8))
    row0 = list(rows[0]) if len(rows) else [1.0] * 8
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = OracleModel(times, lh, row0, float(split), mis, pus, cpfit=flags.get("cpfit", False),
                        trueEPS=flags.get("true_eps", False), smooth=flags.get("smooth", False),
                        unfolded=flags.get("unfolded", False), sampleDate=sample_date, mixtureTH=mixture_th)
        m.jafs_likelihood(mu)
    if m.status != 0:
        return np.full(len(rows), -np.inf), None, m.status, 0.0
    llk = np.array([m.llk_for([float(v) for v in r[1:]], llh_const(r, m.unfolded)) for r in rows])
    return llk, np.array(m.JAFS), 0, float(getattr(m, "max_rate_x_len", 0.0))


def oracle_truth_spectrum(times, lh, split, bands, pulses, sample_date):
    flags = dict(cpfit=True, true_eps=True, smooth=False, unfolded=True)
    _, jafs, st, _ = oracle_eval(times, lh, bands, pulses, flags, sample_date, split, None, None)
    if st != 0:
        raise RuntimeError("oracle truth spectrum failed")
    return [float(v) for v in jafs]


def _worker(args):
    (times, lh, bands, pulses, flags, sample_date, jsfs), chunk = args
    t0 = time.perf_counter()
    try:                                   # one BLAS thread per process, as MiSTI.py:23-25 forces
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(1)
    except Exception:                      # pragma: no cover
        import contextlib
        ctx = contextlib.nullcontext()
    with ctx:
        out = [oracle_eval(times, lh, bands, pulses, flags, sample_date, s, p, jsfs) for s, p in chunk]
    return out, time.perf_counter() - t0


def _children(jobs):
    """_worker(job) for every job, each in a CHILD INTERPRETER started with subprocess (`python -m oracle.batch`), job and result
    pickled over its pipes.  Never a fork: the caller may be a process with a live ROCm runtime (the -m gpu tests), whose forked
    children would inherit the runtime's locks and mapped queues (bench.py starts its baseline as a fresh child for the
    same reason) - and, unlike multiprocessing's "spawn", nothing here depends on the caller's __main__ being importable."""
    import os
    import pickle
    import subprocess
    import sys
    import threading
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), PYTHONDONTWRITEBYTECODE="1")
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.batch"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, cwd=root, env=env) for _ in jobs]
    out = [None] * len(jobs)

    def talk(k):                           # one thread per child: communicate() feeds stdin and drains stdout without deadlock
        data, _ = procs[k].communicate(pickle.dumps(jobs[k]))
        out[k] = data

    threads = [threading.Thread(target=talk, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k, p in enumerate(procs):
        if p.returncode != 0:
            raise RuntimeError("oracle child %d exited with status %s" % (k, p.returncode))
    return [pickle.loads(o) for o in out]


def oracle_batch(workload, indices, processes=1):
    """Oracle on a subset of a Workload's candidates.

    Returns (llk[n][R], status[n], seconds wall); ``oracle_batch.last_runaway`` holds, per
    candidate, the largest corrected rate x interval length (see OracleModel.max_rate_x_len)."""
    import multiprocessing as mp
    w = workload
    common = (w.times, w.lh, w.bands, w.pulses, w.flags, w.sample_date, w.jsfs)
    cands = [(float(w.split_time[i]), None if w.params is None else list(w.params[i])) for i in indices]
    t0 = time.perf_counter()
    if processes <= 1:
        res = [_worker((common, cands))]
    else:
        chunks = [cands[k::processes] for k in range(processes)]
        res = _children([(common, c) for c in chunks])
        # undo the striding
        merged = [None] * len(cands)
        for k, (out, _) in enumerate(res):
            for j, o in enumerate(out):
                merged[k + j * processes] = o
        res = [(merged, 0.0)]
    wall = time.perf_counter() - t0
    out = res[0][0]
    llk = np.array([o[0] for o in out])
    status = np.array([o[2] for o in out], dtype=np.int32)
    oracle_batch.last_runaway = np.array([o[3] for o in out])
    oracle_batch.last_jafs = [o[1] for o in out]
    return llk, status, wall


if __name__ == "__main__":                 # a child of _children(): one pickled job on stdin, its pickled result on stdout
    import pickle
    import sys
    job = pickle.load(sys.stdin.buffer)
    result = _worker(job)
    sys.stdout.buffer.write(pickle.dumps(result))
    sys.stdout.buffer.flush()
