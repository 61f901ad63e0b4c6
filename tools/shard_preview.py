#!/usr/bin/env python3
"""What ONE rank of an N-rank strong-scaling run of a BASELINE workload has to do, measured on one GPU (run on the GPU box): the
shard of rank 0 when whole chains are dealt to the ranks (misti_amd.dist.chain_shards) and when candidates are interleaved
(SURVEY 8e), as single-batch latency.  The ratio to the N = 1 row is the speed-up an N-GPU node can reach at best (the gather of
llk, tens of KB, comes on top).  No multi-GPU hardware is involved: this is a per-rank cost model, not a scaling curve.

    python tools/shard_preview.py [config5] > profiles/rNN_shard_preview.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                      # noqa: E402
from misti_amd import workloads                                   # noqa: E402
from misti_amd.dist import chain_shards, shard_indices            # noqa: E402
from misti_amd.engine import Engine, truth_spectrum               # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
w = workloads.BUILDERS[name](lambda *a: truth_spectrum(*a))
dev = torch.device("cuda", 0)
print("# %s: %d candidates, %d chains" % (w.name, w.n_cand, len({tuple(p) for p in w.params}) if w.params is not None else 1))
print("%-4s %-12s %10s %8s %12s %10s %10s" % ("N", "sharding", "candidates", "chains", "ms per batch", "kernel 1", "kernel 2"))
base = None
for world in (1, 2, 4, 8):
    for how in ("chain", "interleave"):
        if world == 1 and how == "interleave":
            continue
        mine = chain_shards(w.params, w.n_cand, world)[0] if how == "chain" else shard_indices(w.n_cand, 0, world, interleave=True)
        n = len(mine)
        chains = len({tuple(p) for p in w.params[mine]}) if w.params is not None else 1
        with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
            d_split = torch.as_tensor(w.split_time[mine], device=dev)
            d_par = torch.as_tensor(w.params[mine], device=dev).contiguous() if w.n_param else None
            d_j = torch.as_tensor(w.jsfs, device=dev).contiguous()
            R = w.jsfs.shape[0]
            llk = torch.empty((n, R), dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            step = lambda: e.evaluate_dev(n, d_split.data_ptr(), d_par.data_ptr() if d_par is not None else 0, R, d_j.data_ptr(), llk.data_ptr(), 0, 0, 0, 0)
            for _ in range(3):
                step(); e.sync()
            e.enable_timing(True); e.kernel_times(reset=True)
            t0 = time.perf_counter()
            for _ in range(10):
                step(); e.sync()
            dt = (time.perf_counter() - t0) / 10
            ms, cnt = e.kernel_times(reset=True)
        base = base or dt
        print("%-4d %-12s %10d %8d %12.3f %10.3f %10.3f   x%.2f" % (world, how, n, chains, 1e3 * dt, ms["correct"] / max(cnt["correct"], 1),
                                                                    ms["spectrum"] / max(cnt["spectrum"], 1), base / dt))
