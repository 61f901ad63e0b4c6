#!/usr/bin/env python3
"""Lifetime stress of the C ABI on the GPU box: the three suspects of the round-5 driver crash, each by itself,
thousands of times in ONE process, with the crash reporter of tests/conftest.py installed (tests/crashname.c: a signal ends the log
with the native backtrace and the phase / iteration that was running).

    python tools/stress_lifecycle.py [--engine 10000] [--two-phase 300] [--multi 2000] [--lanes 60] [--timer-thread]

  engine     N x (misti_create -> misti_eval_batch on a random small model, as tests/test_gpu_campaign.py makes them -> misti_destroy);
             every result is compared with the FIRST evaluation of the same model (bit for bit): a use-after-free that does not crash shows as a difference
  two-phase  N x (create -> a packed batch that yields chains, i.e. side stream + events, twice -> destroy)
  multi      N x (misti_create_multi on devices [0, 0] -> misti_multi_eval_batch -> misti_destroy_multi): persistent worker threads
  lanes      N x (LanePool of 8 contexts -> 16 batches in flight -> close)
  --timer-thread  a Python thread that wakes every millisecond and allocates (what pytest-timeout's timer thread and the garbage
             collector do to the main thread's ctypes calls)

Prints one line per phase; exit status 0 only if every phase ran to its end with identical results."""
import argparse
import ctypes
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def install_crash_reporter():
    import conftest
    lib = conftest._crash_reporter()
    if lib is not None:
        lib.crashname_install(os.dup(2))
    import faulthandler
    faulthandler.enable()
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", type=int, default=10000)
    ap.add_argument("--two-phase", type=int, default=300)
    ap.add_argument("--multi", type=int, default=2000)
    ap.add_argument("--lanes", type=int, default=60)
    ap.add_argument("--timer-thread", action="store_true")
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    crash = install_crash_reporter()

    def phase(name, i):
        if crash is not None:
            crash.crashname_set(("tools/stress_lifecycle.py phase %s iteration %d" % (name, i)).encode())

    import random_campaign as rc
    from misti_amd import workloads
    from misti_amd.engine import Engine, MultiEngine, truth_spectrum

    stop = threading.Event()
    if a.timer_thread:
        def churn():
            junk = []
            while not stop.is_set():
                junk.append([object() for _ in range(200)])
                if len(junk) > 50:
                    junk.clear()
                time.sleep(0.001)
        threading.Thread(target=churn, daemon=True).start()

    ok = True
    rng = np.random.default_rng(a.seed)
    models = [rc.random_batch(rng) for _ in range(200)]

    def engine_of(c, cls=Engine, **kw):
        return cls(c["times"], c["lh"], c["bands"], c["pulses"], n_param=c["P"], sample_date=c["sd"], **c["flags"], **kw)

    # ---- engine ---------------------------------------------------------------------------------------------------------------
    t0 = time.time()
    first = {}
    diff = 0
    for i in range(a.engine):
        phase("engine", i)
        k = i % len(models)
        c = models[k]
        with engine_of(c) as e:
            r = e.evaluate(c["split"], c["params"], [c["sfs"]])
        key = (r.llk.tobytes(), r.status.tobytes(), r.jafs.tobytes())
        if k not in first:
            first[k] = key
        elif first[k] != key:
            diff += 1
    print("engine    : %d create/evaluate/destroy cycles in %.1f s, %d results differ from the first evaluation of their model" % (a.engine, time.time() - t0, diff), flush=True)
    ok = ok and diff == 0

    # ---- two-phase ------------------------------------------------------------------------------------------------------------
    if a.two_phase:
        t0 = time.time()
        w = workloads.config3(lambda *x: truth_spectrum(*x), n_start=2048)
        want = None
        diff = 0
        for i in range(a.two_phase):
            phase("two-phase", i)
            with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
                r1 = e.evaluate(w.split_time, w.params, w.jsfs)
                r2 = e.evaluate(w.split_time, w.params, w.jsfs)
            key = (r1.llk.tobytes(), r1.status.tobytes())
            if want is None:
                want = key
            if key != want or (r2.llk.tobytes(), r2.status.tobytes()) != want:
                diff += 1
        print("two-phase : %d contexts x 2 packed batches of %d candidates in %.1f s, %d differ" % (a.two_phase, w.n_cand, time.time() - t0, diff), flush=True)
        ok = ok and diff == 0

    # ---- multi ----------------------------------------------------------------------------------------------------------------
    if a.multi:
        t0 = time.time()
        diff = 0
        for i in range(a.multi):
            phase("multi", i)
            k = i % len(models)
            c = models[k]
            with engine_of(c, MultiEngine, devices=(0, 0)) as m:
                r = m.evaluate(c["split"], c["params"], [c["sfs"]])
                if i % 3 == 0:
                    r = m.evaluate(c["split"], c["params"], [c["sfs"]])
            key = (r.llk.tobytes(), r.status.tobytes(), r.jafs.tobytes())
            if k in first and first[k] != key:
                diff += 1
        print("multi     : %d create_multi([0,0])/evaluate/destroy cycles in %.1f s, %d differ from the single-context result" % (a.multi, time.time() - t0, diff), flush=True)
        ok = ok and diff == 0

    # ---- lanes ----------------------------------------------------------------------------------------------------------------
    if a.lanes:
        from misti_amd.lanes import LanePool
        t0 = time.time()
        w = workloads.config2(lambda *x: truth_spectrum(*x), n_split=16, n_rate=16, first_split=56)
        want = None
        diff = 0
        for i in range(a.lanes):
            phase("lanes", i)
            with LanePool(w.times, w.lh, lanes=8, **w.engine_kwargs()) as pool:
                res = pool.map([(w.split_time, w.params, w.jsfs)] * 16)
            keys = {(llk.tobytes(), status.tobytes()) for llk, jafs, status in res}
            if want is None:
                want = next(iter(keys))
            if keys != {want}:
                diff += 1
        print("lanes     : %d pools x 8 contexts x 16 batches in %.1f s, %d differ" % (a.lanes, time.time() - t0, diff), flush=True)
        ok = ok and diff == 0

    stop.set()
    print("stress ok" if ok else "stress FAILED (results differ)", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
