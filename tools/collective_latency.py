#!/usr/bin/env python3
"""Where do the occasional 40 - 50 ms collectives of the single-rank RCCL rehearsal come from?  (run on the GPU box)

    python tools/collective_latency.py [--iters 2000] [--bytes 262144] [--with-engine]

One rank, process group "nccl" (= RCCL), `all_gather_into_tensor` of a `--bytes` tensor issued `--iters` times on a side stream, each
bracketed by HIP events and host timers; optionally with engine batches in flight on other streams (the bench's situation).  Prints the
distribution of device and host durations and, for every collective over 5 ms, its index and the wall-clock gap to the previous outlier - a
period points at a timer (c10d's watchdog / heartbeat threads), random positions at queue scheduling."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--bytes", type=int, default=262144)
    ap.add_argument("--with-engine", action="store_true")
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    n = a.bytes // 8
    src = torch.zeros(n, dtype=torch.float64, device=dev)
    dst = torch.empty(n, dtype=torch.float64, device=dev)
    side = torch.cuda.Stream(device=dev)
    lanes = None
    if a.with_engine:
        from misti_amd import workloads
        from misti_amd.engine import truth_spectrum
        from misti_amd.lanes import LanePool
        w = workloads.config2(lambda *x: truth_spectrum(*x))
        lanes = LanePool(w.times, w.lh, lanes=8, **w.engine_kwargs())
        job = [(w.split_time, w.params, w.jsfs)] * 8
    for _ in range(20):
        with torch.cuda.stream(side):
            dist.all_gather_into_tensor(dst, src)
    torch.cuda.synchronize()
    dev_ms, host_ms, stamp = [], [], []
    t00 = time.perf_counter()
    for i in range(a.iters):
        if lanes is not None and i % 16 == 0:
            lanes.map(job)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            e0.record()
            dist.all_gather_into_tensor(dst, src)
            e1.record()
        e1.synchronize()
        host_ms.append(1e3 * (time.perf_counter() - t0))
        dev_ms.append(e0.elapsed_time(e1))
        stamp.append(time.perf_counter() - t00)
    dev_ms, host_ms, stamp = np.array(dev_ms), np.array(host_ms), np.array(stamp)
    q = [0.5, 0.9, 0.99, 0.999, 1.0]
    print("all_gather_into_tensor of %d bytes, world 1, %d times%s" % (a.bytes, a.iters, ", engine batches in flight" if lanes else ""))
    print("  device ms (HIP events) quantiles 50/90/99/99.9/max: " + " ".join("%.3f" % v for v in np.quantile(dev_ms, q)))
    print("  host   ms (issue + wait) quantiles                 : " + " ".join("%.3f" % v for v in np.quantile(host_ms, q)))
    out = np.nonzero(host_ms > 5.0)[0]
    print("  collectives over 5 ms: %d of %d" % (len(out), a.iters))
    prev = None
    for i in out[:40]:
        print("    #%d at %.3f s: host %.1f ms, device %.3f ms%s" % (i, stamp[i], host_ms[i], dev_ms[i], "" if prev is None else "  (%.3f s after the previous one)" % (stamp[i] - prev)))
        prev = stamp[i]
    if lanes is not None:
        lanes.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
