#!/usr/bin/env python3
"""Per-chain work of the correction kernel on the headline grid (run on the GPU box).

    python tools/stamp_run.py                                                   # counters: batches, speculative steps, max nfev, SVD steps (dense evaluations
                                                                                # and series terms too with a -DMISTI_WORK_COUNTERS=1 build)
    python -m misti_amd.build --out /tmp/stamp.so -DMISTI_STAMP
    MISTI_LIB_AB=1 MISTI_LIB=/tmp/stamp.so python tools/stamp_run.py                            # cycle stamps per phase (profiles/rNN_stamp_longest_chain.txt)

The kernel writes its per-chain counters (or, in the stamped build, clock64() differences per phase) into the last row of
the `pr` output."""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from misti_amd import workloads
from misti_amd.engine import Engine, truth_spectrum
WL = next((a for a in sys.argv[1:] if not a.startswith("-")), "config2")        # config2 | config3 | config5
w = getattr(workloads, WL)(lambda *a: truth_spectrum(*a))
with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
    r = e.evaluate(w.split_time, w.params, w.jsfs, want_pr=True)
    work = r.pr[:, -1, :]         # per candidate: its chain's counters
    # unique chains by rate
    rows = {}
    for k in range(w.n_cand):
        rows[tuple(float(v) for v in w.params[k])] = work[k]
    stamped = float(np.max(work[:, 4])) > 1e5          # cycle stamps, not counters
    tab = sorted(rows.items(), key=lambda kv: -(float(np.sum(kv[1])) if stamped else kv[1][0]))
    print("rate, c_tree, c_collect+c_update, c_next, c_adv, c_batch, c_book  (cycles; STAMP build; plain build: evals dense terms spec max_nfev lm)")
    for rate, wk in tab[:6] + tab[-3:]:
        print(" ".join("%.4g" % v for v in rate), " ".join("%.0f" % v for v in wk), ("total %.3g Mcycles" % (np.sum(wk) / 1e6)) if stamped else "", " per-pass: adv %.0f batch %.0f book %.0f" % tuple(wk[3:6] / max(wk[0], 1)))
    # timing serial
    dev = torch.device("cuda", 0)
    d_split = torch.as_tensor(w.split_time, device=dev); d_par = torch.as_tensor(w.params, device=dev).contiguous(); d_j = torch.as_tensor(w.jsfs, device=dev).contiguous()
    out = torch.empty((w.n_cand, 1), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        e.evaluate_dev(w.n_cand, d_split.data_ptr(), d_par.data_ptr(), 1, d_j.data_ptr(), out.data_ptr()); e.sync()
    e.enable_timing(True); e.kernel_times(reset=True)
    t0 = time.perf_counter()
    for _ in range(10):
        e.evaluate_dev(w.n_cand, d_split.data_ptr(), d_par.data_ptr(), 1, d_j.data_ptr(), out.data_ptr()); e.sync()
    dt = (time.perf_counter() - t0) / 10
    ms, n = e.kernel_times(reset=True)
    print("serial ms/step %.3f" % (dt * 1e3), {k: ms[k] / max(n[k], 1) for k in ms})
