#!/usr/bin/env python3
"""Per-batch and per-kernel times of the default-fit variants beside the --cpfit headline (run on the GPU box).

    python tools/time_default_fit.py profiles/rNN_default_fit_timings.json"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from misti_amd import workloads
from misti_amd.engine import Engine, truth_spectrum
spec = lambda *a: truth_spectrum(*a)
out = {}
def run(name, w, flags=None):
    kw = w.engine_kwargs()
    if flags: kw.update(flags)
    dev = torch.device("cuda", 0)
    with Engine(w.times, w.lh, **kw) as e:
        d_split = torch.as_tensor(w.split_time, device=dev)
        d_par = torch.as_tensor(w.params, device=dev).contiguous() if w.n_param else None
        d_j = torch.as_tensor(w.jsfs, device=dev).contiguous()
        R = w.jsfs.shape[0]
        llk = torch.empty((w.n_cand, R), dtype=torch.float64, device=dev)
        st = torch.empty(w.n_cand, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        step = lambda: e.evaluate_dev(w.n_cand, d_split.data_ptr(), d_par.data_ptr() if d_par is not None else 0, R, d_j.data_ptr(), llk.data_ptr(), 0, 0, 0, st.data_ptr())
        for _ in range(3): step(); e.sync()
        e.enable_timing(True); e.kernel_times(reset=True)
        t0 = time.perf_counter()
        for _ in range(10): step(); e.sync()
        dt = (time.perf_counter() - t0) / 10
        ms, n = e.kernel_times(reset=True)
        ok = float((st.cpu().numpy() == 0).mean())
        import zlib; h = zlib.crc32(llk.cpu().numpy().tobytes())
        out[name] = dict(ms_per_batch=1e3 * dt, kernels={k: ms[k] / max(n[k], 1) for k in ms}, ok=ok, checksum=h)
        print(name, "%.3f ms/batch" % (1e3 * dt), {k: round(ms[k] / max(n[k], 1), 4) for k in ms}, "ok %.3f" % ok, "checksum", h)
run("config2 cpfit", workloads.config2(spec))
run("config2 default fit", workloads.config2(spec), dict(cpfit=False))
run("config4 default fit (256 x 1000)", workloads.config4(spec))
run("config3 default fit, 16384 chains", workloads.config3(spec), dict(cpfit=False))
json.dump(out, open(sys.argv[1], "w"), indent=1) if len(sys.argv) > 1 else None
