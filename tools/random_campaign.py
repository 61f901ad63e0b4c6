#!/usr/bin/env python3
"""Randomised differential campaign (run on the GPU box): random small models, each evaluated as a
BATCH (several split values x parameter vectors, so that chains are shared and the trunk paths run)
through the C ABI and, candidate by candidate, through the oracle.

    python tools/random_campaign.py --make-ref scratch/campaign_ref.json [--models 400] [--seed 1] [--procs 8]   # CPU: oracle
    python tools/random_campaign.py --ref scratch/campaign_ref.json      [--models 400] [--seed 1]                # GPU box

Prints one summary: counts by category and the worst relative errors."""
import argparse
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_batch(rng):
    numT = int(rng.integers(8, 40))
    times = list(np.round(10 ** rng.uniform(-2.3, -0.7, numT - 1), 6))
    # PSMC-like: rates constant in runs of 1-4 intervals, different runs per genome
    lh = np.empty((numT, 2))
    for k in (0, 1):
        t = 0
        while t < numT:
            run = int(rng.integers(1, 5))
            lh[t:t + run, k] = np.round(10 ** rng.uniform(-0.3, 0.4), 4)
            t += run
    sd = int(rng.integers(0, 3)) if rng.random() < 0.3 else 0
    lo = max(2, sd + 1)
    splits = np.sort(rng.choice(np.arange(lo, numT - 2), size=min(int(rng.integers(6, 15)), numT - 2 - lo), replace=False)).astype(float)
    frac = rng.random(len(splits)) < 0.3
    splits[frac] += np.round(rng.uniform(0.1, 0.9, frac.sum()), 3)
    bands, pulses, P = [], [], 0
    first = int(np.floor(splits.min()))
    for pop in (0, 1):
        if rng.random() < 0.7:
            start = int(rng.integers(sd, max(sd + 1, first)))
            end = -1 if rng.random() < 0.6 else int(rng.integers(start + 1, first + 1)) if first > start else -1
            opt = rng.random() < 0.6
            bands.append((pop, start, end, float(np.round(10 ** rng.uniform(-2, 0.0), 4)), P if opt else -1))
            P += int(opt)
    if rng.random() < 0.4 and first - sd >= 1:
        opt = rng.random() < 0.5
        pulses.append((int(rng.integers(0, 2)), int(rng.integers(sd, first)), float(np.round(rng.uniform(0.02, 0.6), 3)), P if opt else -1))
        P += int(opt)
    flags = dict(cpfit=bool(rng.random() < 0.6), true_eps=bool(rng.random() < 0.15), smooth=bool(rng.random() < 0.75),
                 unfolded=bool(rng.random() < 0.5))
    n_par = int(rng.integers(1, 3)) if P else 1
    pars = np.round(10 ** rng.uniform(-2, 0.0, (n_par, max(P, 1))), 4)
    for (pop, t, v, par) in pulses:
        if par >= 0:
            pars[:, par] = np.round(rng.uniform(0.02, 0.6, n_par), 3)
    split = np.repeat(splits, n_par)
    params = np.tile(pars, (len(splits), 1))[:, :P] if P else None
    sfs = [1e5] + [float(v) for v in rng.integers(50, 3000, 7)]
    return dict(times=times, lh=lh.tolist(), sd=sd, split=split, params=params, bands=bands, pulses=pulses, P=P, flags=flags, sfs=sfs)


def oracle_job(args):
    from oracle.batch import oracle_eval
    c, k = args
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        par = list(c["params"][k]) if c["P"] else []
        llk, jafs, st, run = oracle_eval(c["times"], c["lh"], c["bands"], c["pulses"], c["flags"], c["sd"], float(c["split"][k]), par, [c["sfs"]])
    return (None if llk is None else float(llk[0]), None if jafs is None else list(jafs), int(st), float(run))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--make-ref", default="", help="run the oracle (no GPU needed) and write its results here")
    ap.add_argument("--ref", default="", help="oracle results written by --make-ref with the same --models/--seed")
    ap.add_argument("--dump", default="", help="write the HIP path's llk and status of every candidate here (JSON)")
    a = ap.parse_args()
    import json
    import multiprocessing as mp
    from threadpoolctl import threadpool_limits
    rng = np.random.default_rng(a.seed)
    cases = [random_batch(rng) for _ in range(a.models)]
    jobs = [(c, k) for c in cases for k in range(len(c["split"]))]
    if a.make_ref:
        ref = []
        with threadpool_limits(1):
            with mp.get_context("fork").Pool(a.procs) as pool:
                for k, out in enumerate(pool.imap(oracle_job, jobs, chunksize=4)):
                    ref.append(out)
                    if k % 500 == 0:
                        print("oracle %d / %d" % (k, len(jobs)), file=sys.stderr, flush=True)
        if a.make_ref.endswith(".gz"):             # the compact committed fixture: [llk, status, rate x length] per candidate
            import gzip
            with gzip.open(a.make_ref, "wt") as f:
                json.dump({"models": a.models, "seed": a.seed, "n": len(ref), "ref": [[r[0], r[2], r[3]] for r in ref]}, f)
        else:
            json.dump({"models": a.models, "seed": a.seed, "n": len(ref), "ref": ref}, open(a.make_ref, "w"))
        print("wrote", a.make_ref, len(ref), "candidates")
        return
    ref = load_ref(a.ref, a.models, a.seed, len(jobs))
    report = compare(cases, ref, dump=a.dump)
    import gzip
    report["protocol"] = json.load(gzip.open(a.ref, "rt")).get("protocol") if a.ref.endswith(".gz") else None
    print("models %d  seed %d" % (a.models, a.seed))
    print_report(report)


def load_ref(path, models, seed, n_jobs):
    """Oracle results: the full file of --make-ref or the compact committed fixture.  Rows become
    (llk, jafs or None, status, rate x length, spread, perturbed runs without a value, kinds, internal spread, internal
    runs without a value, internal runs): None where tools/self_perturbation.py has not studied the candidate."""
    import gzip
    import json
    d = json.load(gzip.open(path, "rt") if path.endswith(".gz") else open(path))
    assert d["models"] == models and d["seed"] == seed and d["n"] == n_jobs, "reference file made with other settings"
    out = []
    for r in d["ref"]:
        if len(r) == 4 and (r[1] is None or isinstance(r[1], list)):      # --make-ref: llk, jafs, status, run
            out.append((r[0], r[1], r[2], r[3], None, None, None, None, None, None))
        else:                                                               # fixture: llk, status, run [, spread, nfail, kinds]
            ext = tuple(r[3:6]) if len(r) >= 6 else (None, None, None)
            ext += tuple(r[6:9]) if len(r) >= 9 else (None, None, None)      # internal spread, its failed runs, its runs
            out.append((r[0], None, r[1], r[2]) + ext)
    return out


def compare(cases, ref, dump=""):
    """Evaluate every model as one batch through the C ABI and compare candidate by candidate under the contract of
    tests/parity.py: |llk - ref| <= 1e-9 |ref| + rounding floor, or <= 10 x the reference's own measured spread for
    that candidate (under 2^-48 input perturbations, or under one ulp in its own pair-chain expm); a failure against a
    value (either way) only where the reference itself flips under perturbation."""
    from parity import SELF_FACTOR, llk_tol
    from misti_amd.engine import Engine
    stats = dict(candidates=0, both_fail=0, status_mismatch=0, status_flip_ok=0, tight=0, self_bound=0, internal_bound=0, outside=0, unstudied=0)
    worst_tight = worst_factor = worst_internal = 0.0
    bad, outside, factors, ifactors, dumped = [], [], [], [], []
    pos = 0
    for ci, c in enumerate(cases):
        n = len(c["split"])
        with Engine(c["times"], c["lh"], c["bands"], c["pulses"], n_param=c["P"], sample_date=c["sd"], **c["flags"]) as e:
            r = e.evaluate(c["split"], c["params"], [c["sfs"]])
        for k in range(n):
            o_llk, o_jafs, o_st, run, spread, nfail, kinds, internal, ifail, iruns = ref[pos + k]
            stats["candidates"] += 1
            dumped.append([float(r.llk[k, 0]) if np.isfinite(r.llk[k, 0]) else None, int(r.status[k])])
            if o_st != 0 or r.status[k] != 0:
                if o_st != 0 and r.status[k] != 0:
                    stats["both_fail"] += 1
                elif (kinds and ((o_st != 0 and nfail < kinds) or (o_st == 0 and nfail > 0))) or (o_st == 0 and ifail):
                    stats["status_flip_ok"] += 1          # the reference itself flips between a value and a failure
                else:
                    stats["status_mismatch"] += 1
                    bad.append(("status", pos + k, ci, k, o_st, int(r.status[k]), float(c["split"][k]), run, kinds))
                continue
            err = abs(r.llk[k, 0] - o_llk)
            rel = err / abs(o_llk)
            if err <= llk_tol(o_llk, c["sfs"], o_jafs if o_jafs is not None else r.jafs[k], c["flags"]["unfolded"]):
                stats["tight"] += 1
                worst_tight = max(worst_tight, rel)
            elif spread is not None and spread > 0 and rel <= SELF_FACTOR * spread:
                stats["self_bound"] += 1
                worst_factor = max(worst_factor, rel / spread)
                factors.append((rel / spread, rel, spread, pos + k, ci, k))
            elif internal is not None and internal > 0 and rel <= SELF_FACTOR * internal:
                stats["internal_bound"] += 1
                worst_internal = max(worst_internal, rel / internal)
                ifactors.append((rel / internal, rel, internal, spread, pos + k, ci, k))
            else:
                stats["outside"] += 1
                stats["unstudied"] += spread is None
                outside.append((rel, spread, pos + k, ci, k, float(c["split"][k]), run, c["flags"]["cpfit"], kinds, internal))
        pos += n
    if dump:
        import json
        json.dump({"n": len(dumped), "hip": dumped}, open(dump, "w"))
    outside.sort(key=lambda b: -b[0])
    factors.sort(reverse=True)
    ifactors.sort(reverse=True)
    return dict(stats=stats, worst_tight=worst_tight, worst_factor=worst_factor, worst_internal=worst_internal, bad=bad, outside=outside,
                factors=factors, ifactors=ifactors)


def print_report(rep):
    from parity import SELF_FACTOR
    stats = rep["stats"]
    print("# reference VALUES of this report: the NumPy/SciPy ORACLE (oracle/misti_oracle.py: the reference's own SciPy calls; agrees with the reference to")
    print("# <= 1e-12, bit for bit on almost every case, on the reference-generated goldens), not /root/reference itself.  SPREADS: the fixture's protocol -")
    print("# " + (rep.get("protocol") or "round 3: the oracle under 4-64 input perturbations, deepened for single candidates"))
    print(stats)
    print("within 1e-9 (+ rounding floor): %d, worst %.3g" % (stats["tight"], rep["worst_tight"]))
    print("within %g x the reference's own measured spread under input perturbations: %d, worst factor %.2f" % (SELF_FACTOR, stats["self_bound"], rep["worst_factor"]))
    if rep["factors"]:
        fs = np.array([f[0] for f in rep["factors"]])
        print("   distribution of the factor |llk - ref| / spread over those %d: <= 1: %d (%.0f %%), <= 3: %d (%.0f %%); median %.2f"
              % (len(fs), (fs <= 1).sum(), 100.0 * (fs <= 1).mean(), (fs <= 3).sum(), 100.0 * (fs <= 3).mean(), float(np.median(fs))))
    for f in rep["factors"][:8]:
        print("   factor %.2f  rel %.3g  spread %.3g  candidate %d (model %d cand %d)" % f)
    print("within %g x its spread under one ulp in its own pair-chain expm: %d more, worst factor %.2f" % (SELF_FACTOR, stats["internal_bound"], rep["worst_internal"]))
    for f in rep["ifactors"][:12]:
        print("   factor %.2f  rel %.3g  internal %.3g  (input spread %s)  candidate %d (model %d cand %d)" % f)
    print("OUTSIDE the contract in this FIRST PASS: %d (%d of them without a first-pass spread); every one is then run through /root/reference itself"
          % (stats["outside"], stats["unstudied"]))
    print("   (tests/golden/golden_campaign.json; tests/test_gpu_golden.py::test_campaign_worst holds each to the reference's own value and spreads)")
    for b in rep["outside"][:40]:
        print("   rel %.3g  spread %s  candidate %d (model %d cand %d) split %.3f  rate x len %.3g  cpfit %s  kinds %s  internal %s" % b)
    print("status: %d both fail, %d reference flips under perturbation, %d MISMATCHES" % (stats["both_fail"], stats["status_flip_ok"], stats["status_mismatch"]))
    for b in rep["bad"][:20]:
        print("  ", b)


if __name__ == "__main__":
    main()
