#!/usr/bin/env python3
"""Where one burst of K batches on K lanes goes (the driver's shape of bench.py: --steps 20 on 20 lanes from an idle device).

rocprofv3's kernel trace serialises the dispatches of a process, so it cannot show batches that overlap.  This script stamps every
lane's stream with HIP events instead - one before and one behind each batch - plus the library's own per-kernel event pairs
(misti_enable_timing), all read against ONE event recorded before the first launch:

    python tools/burst_timeline.py [--lanes 20] [--steps 20] [--reps 5]

prints, per lane of the best repetition: when its batch was issued (host clock), when its first kernel could start and when its last
kernel ended (device clock), and the host's wall time from the first launch to the return of torch.cuda.synchronize()."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "22")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=20)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--stage-timing", action="store_true", help="also the library's per-stage event pairs (two more events per stage and batch)")
    a = ap.parse_args()
    import torch
    from misti_amd import workloads
    from misti_amd.engine import Lanes, truth_spectrum
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = workloads.BUILDERS[a.workload](lambda *x: truth_spectrum(*x, device=0))
    n, R, P = w.n_cand, int(w.jsfs.shape[0]), w.n_param
    d_split = torch.as_tensor(w.split_time, dtype=torch.float64, device=dev)
    d_par = torch.as_tensor(w.params, dtype=torch.float64, device=dev).contiguous() if P else None
    d_jsfs = torch.as_tensor(w.jsfs, dtype=torch.float64, device=dev).contiguous()
    pool = Lanes(w.times, w.lh, device=0, lanes=a.lanes, **w.engine_kwargs())
    pool.set_hints(integer_splits=bool(np.all(w.split_time == np.floor(w.split_time))))
    L = pool.n_lanes
    eng = [pool.engine(i) for i in range(L)]
    streams = [torch.cuda.ExternalStream(e.stream_handle(), device=dev) for e in eng]
    llk = [torch.empty((n, R), dtype=torch.float64, device=dev) for _ in range(L)]
    jafs = [torch.empty((n, 7), dtype=torch.float64, device=dev) for _ in range(L)]
    status = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(L)]
    torch.cuda.synchronize()

    def step(i):
        pool.evaluate_dev(i, n, d_split.data_ptr(), d_par.data_ptr() if P else 0, R, d_jsfs.data_ptr(), llk[i].data_ptr(), jafs[i].data_ptr(), 0, 0, status[i].data_ptr())

    for i in range(L):                      # first batch of every context: allocations, launch-shape hint
        step(i)
    torch.cuda.synchronize()
    for i in range(min(5, L)):
        step(i)
    torch.cuda.synchronize()
    if a.stage_timing:
        for e in eng:
            e.enable_timing(True)
            e.kernel_times(reset=True)
    best = None
    for rep in range(a.reps):
        K = a.steps
        ev0 = torch.cuda.Event(enable_timing=True)
        before = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
        behind = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
        torch.cuda.synchronize()
        time.sleep(0.01)                     # an idle device, as between the driver's regions
        t0 = time.perf_counter()
        ev0.record(streams[0])
        issued = []
        for k in range(K):
            i = k % L
            before[k].record(streams[i])
            step(i)
            behind[k].record(streams[i])
            issued.append(time.perf_counter() - t0)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        rec = dict(wall=wall * 1e3, issue=t_issue * 1e3, issued=[x * 1e3 for x in issued],
                   start=[ev0.elapsed_time(b) for b in before], end=[ev0.elapsed_time(b) for b in behind])
        if a.stage_timing:
            rec["stages"] = [e.kernel_times(reset=True)[0] for e in eng]
        if best is None or rec["wall"] < best["wall"]:
            best = rec
    print("# %s: %d steps on %d lanes, best of %d repetitions; ms from the first launch" % (a.workload, a.steps, L, a.reps))
    print("# wall (first launch -> torch.cuda.synchronize returns) %.3f ms = %.3e evals/s; host issue %.3f ms; last batch ends (device) %.3f ms -> the tail behind it %.3f ms"
          % (best["wall"], a.steps * n * R / best["wall"] * 1e3, best["issue"], max(best["end"]), best["wall"] - max(best["end"])))
    print("# step lane issued start end duration" + ("  chains+setup spectrum" if a.stage_timing else ""))
    for k in range(a.steps):
        extra = ""
        if a.stage_timing and k < L:
            st = best["stages"][k % L]
            extra = "  %.3f %.3f" % (st["correct"], st["spectrum"])
        print("%2d %2d %.3f %.3f %.3f %.3f%s" % (k, k % L, best["issued"][k], best["start"][k], best["end"][k], best["end"][k] - best["start"][k], extra))
    pool.close()


if __name__ == "__main__":
    main()
