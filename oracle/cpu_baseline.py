"""ctypes front of the compiled CPU baseline (oracle/cpu/misti_cpu.cpp; TEST INFRASTRUCTURE: used by tests/ and the
cpu_baseline leg of bench.py only).  C++17 + OpenMP restatement of the reference's algorithm - dense expm + inverse per
interval, SciPy's trust-region solver restated - one candidate per OpenMP task."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "cpu", "libmisti_cpu.so")


class Band(C.Structure):
    _fields_ = [("pop", C.c_int32), ("start", C.c_int32), ("end", C.c_int32), ("param", C.c_int32), ("value", C.c_double)]


class Pulse(C.Structure):
    _fields_ = [("pop", C.c_int32), ("time", C.c_int32), ("param", C.c_int32), ("pad", C.c_int32), ("value", C.c_double)]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
        _lib = C.CDLL(LIB)
        _lib.misti_cpu_eval.restype = C.c_int
    return _lib


def cpu_eval(times, lh, bands, pulses, flags, sample_date, split_time, params, jsfs_rows, n_param, mixture_th=0.0, threads=1):
    """Candidates x replicates on the host cores: (llk[n][R], jafs[n][7], status[n], rate_x_len[n], seconds)."""
    lib = load()
    times = np.ascontiguousarray(times, dtype=np.float64)
    lh = np.ascontiguousarray(lh, dtype=np.float64)
    numT = lh.shape[0]
    split = np.ascontiguousarray(np.atleast_1d(split_time), dtype=np.float64)
    n = split.shape[0]
    par = np.ascontiguousarray(params, dtype=np.float64).reshape(n, n_param) if n_param else np.zeros((n, 0))
    rows = np.ascontiguousarray(jsfs_rows, dtype=np.float64).reshape(-1, 8)
    R = rows.shape[0]
    b = (Band * max(1, len(bands)))(*[Band(int(p), int(s), int(e), int(q), float(v)) for p, s, e, v, q in bands])
    u = (Pulse * max(1, len(pulses)))(*[Pulse(int(p), int(t), int(q), 0, float(v)) for p, t, v, q in pulses])
    fl = (1 if flags.get("cpfit") else 0) | (2 if flags.get("true_eps") else 0) | (4 if flags.get("smooth") else 0) | (8 if flags.get("unfolded") else 0)
    llk = np.empty((n, R))
    jafs = np.empty((n, 7))
    status = np.empty(n, dtype=np.int32)
    run = np.empty(n)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    t0 = time.perf_counter()
    lib.misti_cpu_eval(C.c_int(numT), C.c_int(int(sample_date)), C.c_uint(fl), C.c_double(float(mixture_th)), ptr(times), ptr(lh),
                       C.c_int(len(bands)), b, C.c_int(len(pulses)), u, C.c_int(int(n_param)), C.c_int64(n), ptr(split), ptr(par),
                       C.c_int64(R), ptr(rows), ptr(llk), ptr(jafs), ptr(status), ptr(run), C.c_int(int(threads)))
    return llk, jafs, status, run, time.perf_counter() - t0
