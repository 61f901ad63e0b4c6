#!/usr/bin/env python3
"""The per-candidate contract of tests/parity.py for EVERY candidate of BASELINE's grids (run on the GPU box):

    python tools/fullsize_report.py [config2 config3 config5 config2:default config3:default ...] [--kinds 8] [--dump gpurun_out]

For each workload: the HIP path's llk / status of every candidate against the compiled CPU baseline (oracle/cpu/misti_cpu.cpp, the
reference's algorithm restated and pinned on the reference-generated goldens) and - for every candidate the HIP path does not match
to 1e-9 - that candidate's own spread under `--kinds` 2^-48 perturbations of the inputs.  One report line per workload, and with
--dump the device's values, the baseline's values and the outlier list as <dump>/fullsize_<name>.npz, so that the build container
can run /root/reference itself on any candidate afterwards (tests/golden/make_fullsize.py) without a GPU.

The checker is NOT /root/reference; the candidates it puts outside the contract are then studied with the reference."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np


def build(spec_name, spectrum):
    from misti_amd import workloads
    name, _, fit = spec_name.partition(":")
    kw = {}
    if fit:
        kw["cpfit"] = fit == "cpfit"
    return getattr(workloads, name)(spectrum, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="*", default=["config2", "config5", "config3"])
    ap.add_argument("--kinds", type=int, default=8)
    ap.add_argument("--internal", type=int, default=8, help="runs with one ulp of noise in the baseline's pair-chain expm, per candidate not within 1e-9")
    ap.add_argument("--stride", type=int, default=1, help="every stride-th candidate (1 = the whole grid)")
    ap.add_argument("--offset", type=int, default=0, help="first candidate of the stride (a grid too long for one GPU call is done in `stride` calls)")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 16)
    ap.add_argument("--dump", default="")
    a = ap.parse_args()
    from parity import baseline_contract
    from misti_amd.engine import Engine, truth_spectrum
    spec = lambda *x: truth_spectrum(*x)
    print("# checker: the compiled CPU baseline (oracle/cpu/misti_cpu.cpp, the reference's algorithm restated, pinned on the reference-generated goldens): values and")
    print("# spreads (%d 2^-48 input perturbations + %d one-ulp-in-expm runs per candidate that is not within 1e-9) - not /root/reference itself; %d host threads"
          % (a.kinds, a.internal, a.threads))
    for spec_name in a.workloads:
        t0 = time.time()
        w = build(spec_name, spec)
        idx = np.arange(a.offset, w.n_cand, a.stride)
        split = w.split_time[idx]
        par = None if w.params is None else w.params[idx]
        with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
            r = e.evaluate(split, par, w.jsfs[:1])
        sub = type(w)(w.name, w.times, w.lh, w.bands, w.pulses, w.n_param, w.flags, w.sample_date, split, par, w.truth, w.jsfs)
        rep = baseline_contract(sub, np.arange(len(idx)), r.llk, r.status, threads=a.threads, kinds=a.kinds, internal=a.internal)
        out = [(int(idx[k]), float(rep["rel"][k]), float(rep["run"][k])) for k in rep["outside"]]
        mis = [(int(idx[k]), int(rep["base_status"][k]), int(r.status[k])) for k in rep["mismatch"]]
        print(spec_name, "every candidate" if a.stride == 1 else "stride %d offset %d" % (a.stride, a.offset), ": n", len(idx), "both", rep["both"], "tight", rep["tight"],
              "frac %.4f" % (rep["tight"] / max(1, rep["both"])), "self", rep["self_bound"], "outside", len(out), out, "mismatch", len(mis), mis,
              "worst_tight %.3g" % rep["worst_tight"], "(%.0f s)" % (time.time() - t0), flush=True)
        if a.dump:
            os.makedirs(a.dump, exist_ok=True)
            tag = spec_name.replace(":", "_") + ("" if a.stride == 1 else "_s%d_o%d" % (a.stride, a.offset))
            np.savez_compressed(os.path.join(a.dump, "fullsize_%s.npz" % tag), idx=idx, hip_llk=r.llk[:, 0], hip_status=r.status, hip_jafs=r.jafs,
                                base_llk=rep["base_llk"], base_status=rep["base_status"], run=rep["run"], rel=rep["rel"],
                                outside=np.array([o[0] for o in out], dtype=np.int64), mismatch=np.array([m[0] for m in mis], dtype=np.int64),
                                jsfs=w.jsfs[0], times=np.array(w.times), lh=np.array(w.lh))
            json.dump({"workload": spec_name, "name": w.name, "n": int(len(idx)), "both": rep["both"], "tight": rep["tight"], "self": rep["self_bound"],
                       "outside": out, "mismatch": mis, "worst_tight": rep["worst_tight"], "kinds": a.kinds, "internal": a.internal},
                      open(os.path.join(a.dump, "fullsize_%s.json" % tag), "w"))


if __name__ == "__main__":
    main()
