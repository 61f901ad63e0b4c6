#!/usr/bin/env python3
"""Reference-run goldens for candidates of BASELINE's FULL-SIZE grids (needs /root/reference: build container only).

The full-size parity tests (tests/test_gpu_fullsize.py, tools/fullsize_report.py) check every candidate of configs 2, 3 and 5 against
the compiled CPU baseline.  The candidates that check puts OUTSIDE the contract, and an evenly spaced sample of the default-fit grids
fixed before any result was looked at, are run here through /root/reference ITSELF:

    python tests/golden/make_fullsize.py values  config5 12315 12667 ...      # one reference run each -> scratch/fullsize_values_<wl>.json
    python tests/golden/make_fullsize.py study   config5 12315 ...            # + 64 input perturbations, 16 one-ulp-in-expm runs, 16
                                                                              #   one-ulp-in-residual runs, solver trace -> golden_fullsize.json
    python tests/golden/make_fullsize.py default                              # 48 candidates of config2 / config3 under the DEFAULT fit, evenly
                                                                              #   spaced, 16 + 16 runs each + traces -> golden_default_fit.json

`values` answers "who is right, the device or the checker" at 1.5 s per candidate; `study` is the depth camp_m148_c12 got in round 3
(tests/golden/internal_noise.py) and decides whether a candidate lies inside clause 2 of the contract BY THE REFERENCE'S OWN MEASUREMENT.
Workloads are rebuilt here with the oracle's truth spectrum; the dump of tools/fullsize_report.py (--dump) carries the data JSFS and rates
the GPU box used, and the two are asserted equal."""
import gzip
import json
import multiprocessing as mp
import os
import sys
import time
import warnings

import numpy

numpy.mat = numpy.asmatrix
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import make_golden as mg               # noqa: E402  (imports the reference)
import internal_noise as inz           # noqa: E402
import parity                          # noqa: E402

N_INPUT, N_INTERNAL, N_RESIDUAL = 64, 16, 16
DUMP = os.environ.get("MISTI_FS_DUMP") or os.path.join(ROOT, "gpurun_out", "fs")


def workload(spec):
    from misti_amd import workloads
    from oracle.batch import oracle_truth_spectrum
    name, _, fit = spec.partition(":")
    kw = {"cpfit": fit == "cpfit"} if fit else {}
    w = getattr(workloads, name)(oracle_truth_spectrum, **kw)
    p = os.path.join(DUMP, "fullsize_%s.npz" % spec.replace(":", "_"))
    dump = numpy.load(p) if os.path.exists(p) else None
    if dump is not None:
        assert numpy.array_equal(dump["jsfs"], w.jsfs[0]) and numpy.array_equal(dump["lh"], numpy.array(w.lh)) and numpy.array_equal(dump["times"], numpy.array(w.times))
    return w, dump


def reference_args(w, idx):
    """Candidate idx of workload w as the reference's constructor sees it."""
    from oracle.batch import _mis_pus
    par = [float(v) for v in w.params[idx]] if w.params is not None else []
    split = float(w.split_time[idx])
    split = int(split) if split == int(split) else split
    end = int(split) + (1 if split % 1 else 0)
    mis, pus = _mis_pus(w.bands, w.pulses, end, par)
    kw = dict(smooth=bool(w.flags["smooth"]))
    if w.flags["cpfit"]:
        kw["cpfit"] = True
    if w.flags["unfolded"]:
        kw["unfolded"] = True
    if w.flags["true_eps"]:
        kw["trueEPS"] = True
    if w.sample_date:
        kw["sampleDate"] = int(w.sample_date)
    order = [b[4] for b in w.bands if b[4] >= 0] + [b[3] for b in w.pulses if b[3] >= 0]
    return list(w.times), [list(x) for x in w.lh], [float(v) for v in w.jsfs[0]], split, mis, pus, kw, [par[i] for i in order]


_W = {}


def _value_job(job):
    spec, idx = job
    w, _ = _W[spec]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = reference_args(w, idx)
        rec = mg.run_reference(*a)
    return idx, rec["llh"]


def _study_job(job):
    spec, idx, n_in, n_int, n_res, trace = job
    w, _ = _W[spec]
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        times, lam, sfs, split, mis, pus, kw, params = reference_args(w, idx)
        c = mg.case("%s_c%d" % (spec.replace(":", "_"), idx), times, lam, sfs, split, mis, pus, params, **kw)
        o = c["out"]
        if o["llh"] is not None:
            _, o["spread"], o["pert_fail"], o["pert_llh"] = mg.perturbation_study(times, lam, sfs, split, mis, pus, kw, params, o["llh"], n_in)
            o["sens"] = None if o["pert_fail"] else o["spread"] / parity.PERTURB
            o["pert_kinds"] = n_in
            i = c["in"]
            vals = [inz.run(i, inz.NoisyLinalg(7000 + s)) for s in range(n_int)]
            fin = [v for v in vals if v is not None]
            o["internal_spread"] = max(abs(v - o["llh"]) / abs(o["llh"]) for v in fin) if fin else None
            o["internal_fail"], o["internal_runs"], o["internal_llh"] = n_int - len(fin), n_int, vals
            if n_res:
                vals = [inz.run(i, None, inz.NoisyOptimize(9000 + s)) for s in range(n_res)]
                fin = [v for v in vals if v is not None]
                o["residual_spread"] = max(abs(v - o["llh"]) / abs(o["llh"]) for v in fin) if fin else None
                o["residual_fail"], o["residual_runs"] = n_res - len(fin), n_res
        else:
            failing_study(c, n_in, n_int)
        o.pop("Pr", None)
        tr = mg.traced(c) if (trace and o["llh"] is not None) else None
    c["fullsize"] = {"workload": spec, "cand": int(idx)}
    c["ref_seconds"] = round(time.time() - t0, 1)
    return c, tr


def failing_study(c, n_in, n_int):
    """A case the reference fails on ("Lambda correction failed"): does it find a value under the same perturbed runs?
    pert_finite / internal_finite = runs with a value (of pert_kinds / internal_runs)."""
    i, o = c["in"], c["out"]
    vals = [mg.run_reference(*parity.perturbed(i["times"], i["lambdas"], k), i["sfs"], i["split"], i["mi"], i["pu"], i["kw"], i["params"])["llh"] for k in range(n_in)]
    o["pert_finite"], o["pert_kinds"], o["pert_llh"] = sum(1 for v in vals if v is not None), n_in, vals
    vals = [inz.run(i, inz.NoisyLinalg(7000 + s)) for s in range(n_int)]
    o["internal_finite"], o["internal_runs"], o["internal_llh"] = sum(1 for v in vals if v is not None), n_int, vals


def _patch_job(c):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        failing_study(c, 16, 16)
    return c


def pool_map(fn, jobs, procs):
    from threadpoolctl import threadpool_limits
    with threadpool_limits(1):
        with mp.get_context("fork").Pool(procs) as pool:
            for k, r in enumerate(pool.imap(fn, jobs)):
                print("  %d / %d" % (k + 1, len(jobs)), file=sys.stderr, flush=True)
                yield r


def write_golden(path, cases, traces, note):
    for c in cases:
        c["in"] = dict(c["in"])
    grids = mg.dedupe(cases)
    json.dump({"generator": "tests/golden/make_fullsize.py", "scipy": "1.15.3", "numpy": "2.2.6", "note": note, "grids": grids, "cases": cases}, open(path, "w"))
    if traces:
        with gzip.open(path.replace(".json", "_traces.json.gz"), "wt") as f:
            json.dump({"generator": "tests/golden/make_fullsize.py", "scipy": "1.15.3", "cases": traces}, f)


def main():
    mode = sys.argv[1]
    procs = int(os.environ.get("PROCS", "6"))
    if mode == "values":
        spec = sys.argv[2]
        idxs = [int(v) for v in sys.argv[3:]]
        _W[spec] = workload(spec)
        _, dump = _W[spec]
        if not idxs:
            idxs = sorted(set(int(v) for v in dump["outside"]) | set(int(v) for v in dump["mismatch"]))
        out = {}
        for idx, llh in pool_map(_value_job, [(spec, i) for i in idxs], procs):
            rec = {"ref": llh}
            if dump is not None:
                h, b = float(dump["hip_llk"][idx]), float(dump["base_llk"][idx])
                rec.update(hip=h if numpy.isfinite(h) else None, base=b if numpy.isfinite(b) else None, run=float(dump["run"][idx]))
                if llh is not None and numpy.isfinite(h):
                    rec["hip_rel"] = abs(h - llh) / abs(llh)
                if llh is not None and numpy.isfinite(b):
                    rec["base_rel"] = abs(b - llh) / abs(llh)
            out[idx] = rec
        os.makedirs(os.path.join(ROOT, "scratch"), exist_ok=True)
        p = os.path.join(ROOT, "scratch", "fullsize_values_%s.json" % spec.replace(":", "_"))
        json.dump(out, open(p, "w"), indent=0)
        n_dev = sum(1 for r in out.values() if r.get("hip_rel", 1) > 1e-9)
        print("%s: %d candidates through the reference; device beyond 1e-9 of the REFERENCE on %d, checker beyond 1e-9 on %d -> %s"
              % (spec, len(out), n_dev, sum(1 for r in out.values() if r.get("base_rel", 0) > 1e-9), p))
        return
    if mode == "study":
        # python make_fullsize.py study spec idx idx ... [spec2 idx ...]
        jobs, spec = [], None
        for a in sys.argv[2:]:
            if a.lstrip("-").isdigit():
                jobs.append((spec, int(a), N_INPUT, N_INTERNAL, N_RESIDUAL, True))
            else:
                spec = a
                if spec not in _W:
                    _W[spec] = workload(spec)
        path = os.path.join(HERE, "golden_fullsize.json")
        cases, traces = [], []
        for c, tr in pool_map(_study_job, jobs, procs):
            cases.append(c)
            if tr:
                traces.append(tr)
        write_golden(path, cases, traces, "candidates of BASELINE's full-size grids that tools/fullsize_report.py (round 4, compiled baseline as checker) put outside the "
                     "contract, run through /root/reference with %d input perturbations, %d one-ulp-in-expm and %d one-ulp-in-residual runs" % (N_INPUT, N_INTERNAL, N_RESIDUAL))
        print("wrote %d cases -> %s" % (len(cases), path))
        return
    if mode == "default":
        jobs = []
        for spec, n in (("config2:default", 4096), ("config3:default", 16384)):
            _W[spec] = workload(spec)
            # 24 per grid, evenly spaced, fixed before any result was looked at; config 2's grid is split-major with 64 rates per split, so a stride
            # that is not a multiple of 64 walks through splits and rates alike
            idxs = [int(round((k + 0.5) * n / 24.0)) for k in range(24)]
            jobs += [(spec, i, 16, 16, 0, True) for i in idxs]
        # ... plus the candidates named on the command line (`default config2:default 545 865 ... config3:default 344 ...`): those the full-grid
        # check against the compiled baseline flagged (outside, or a failure against a value), at the same depth
        spec = None
        for a in sys.argv[2:]:
            if a.lstrip("-").isdigit():
                if (spec, int(a), 16, 16, 0, True) not in jobs:
                    jobs.append((spec, int(a), 16, 16, 0, True))
            else:
                spec = a
        path = os.path.join(HERE, "golden_default_fit.json")
        cases, traces = [], []
        for c, tr in pool_map(_study_job, jobs, procs):
            cases.append(c)
            if tr:
                traces.append(tr)
        write_golden(path, cases, traces, "48 candidates of BASELINE configs 2 and 3 under the reference's DEFAULT fit (MiSTI.py:86,213), evenly spaced (the first 48 "
                     "cases), and the candidates the full-grid check of tools/fullsize_report.py flagged, through /root/reference with 16 input perturbations and "
                     "16 one-ulp-in-expm runs each, with solver traces")
        print("wrote %d cases -> %s" % (len(cases), path))
        return
    if mode == "extra":
        # Round 5: candidates the first pass flags once the contract's factor is 3 (tests/parity.py) that have no study yet, in a file of their
        # own (golden_fullsize_r05.json): `extra config3 13340 config2:default 97 225 ...`; --cpfit workloads at the depth of `study`, default-fit ones
        # at the depth of `default`.
        jobs, spec = [], None
        args = sys.argv[2:]
        out_name = "golden_fullsize_r05.json"
        if args and args[0] == "--out":              # `extra --out golden_config5_default_sample.json config5:default 49080 ...`: the same study into another fixture
            out_name, args = args[1], args[2:]
        for a in args:
            if a.lstrip("-").isdigit():
                deep = not spec.endswith(":default")
                jobs.append((spec, int(a), N_INPUT if deep else 16, N_INTERNAL, N_RESIDUAL if deep else 0, True))
            else:
                spec = a
                if spec not in _W:
                    _W[spec] = workload(spec)
        path = os.path.join(HERE, out_name)
        cases, traces = [], []
        note = ("round 5: candidates of BASELINE's full-size grids the first pass (compiled baseline as checker, factor 3) put outside the contract "
                "and that had no reference-run study yet; --cpfit workloads with 64 + 16 + 16 runs, default-fit workloads with 16 + 16")
        if os.path.exists(path):                    # the file grows: cases of earlier calls stay (grids re-expanded, then deduplicated again on writing)
            d = json.load(open(path))
            note = d.get("note", note)
            for c in d["cases"]:
                if "grid" in c["in"]:
                    g = d["grids"][c["in"].pop("grid")]
                    c["in"]["times"], c["in"]["lambdas"] = g["times"], g["lambdas"]
                cases.append(c)
            tp = path.replace(".json", "_traces.json.gz")
            if os.path.exists(tp):
                traces = json.load(gzip.open(tp, "rt"))["cases"]
            done = {(c["fullsize"]["workload"], c["fullsize"]["cand"]) for c in cases}
            jobs = [j for j in jobs if (j[0], j[1]) not in done]
        for c, tr in pool_map(_study_job, jobs, procs):
            cases.append(c)
            if tr:
                traces.append(tr)
        write_golden(path, cases, traces, note)
        print("wrote %d cases -> %s" % (len(cases), path))
        return
    if mode == "default256":
        # Round 5: default-fit evidence at BASELINE size - 256 candidates FIXED IN ADVANCE, evenly spaced over the grids of
        # configs 2, 3 AND 5 under the reference's default fit (85 + 85 + 86; offset half a stride, as `default` above), each through
        # /root/reference with 16 input perturbations + 16 one-ulp-in-expm runs and its solver trace.  No candidate is chosen by looking at a result.
        jobs = []
        for spec, n, k_n in (("config2:default", 4096, 85), ("config3:default", 16384, 85), ("config5:default", 65536, 86)):
            _W[spec] = workload(spec)
            jobs += [(spec, int(round((k + 0.5) * n / float(k_n))), 16, 16, 0, True) for k in range(k_n)]
        path = os.path.join(HERE, "golden_default_fit_256.json")
        cases, traces = [], []
        for c, tr in pool_map(_study_job, jobs, procs):
            cases.append(c)
            if tr:
                traces.append(tr)
        write_golden(path, cases, traces, "256 candidates of BASELINE configs 2, 3 and 5 under the reference's DEFAULT fit (MiSTI.py:86,213), evenly spaced and fixed in "
                     "advance (85 + 85 + 86), through /root/reference with 16 input perturbations and 16 one-ulp-in-expm runs each, with solver traces")
        print("wrote %d cases -> %s" % (len(cases), path))
        return
    if mode == "patch-failures":
        # golden file written before failing cases got their perturbed runs: add them in place
        path = os.path.join(HERE, sys.argv[2])
        d = json.load(open(path))
        grids = d["grids"]
        todo = []
        for c in d["cases"]:
            if c["out"]["llh"] is None and "internal_finite" not in c["out"]:
                cc = json.loads(json.dumps(c))
                cc["in"]["times"], cc["in"]["lambdas"] = grids[c["in"]["grid"]]["times"], grids[c["in"]["grid"]]["lambdas"]
                todo.append(cc)
        done = {c["name"]: c for c in pool_map(_patch_job, todo, procs)}
        for c in d["cases"]:
            if c["name"] in done:
                c["out"] = done[c["name"]]["out"]
        json.dump(d, open(path, "w"))
        print("patched %d failing cases in %s" % (len(done), path))
        return
    raise SystemExit("mode: values | study | default | default256 | patch-failures")


if __name__ == "__main__":
    main()
