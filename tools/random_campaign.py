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
        json.dump({"models": a.models, "seed": a.seed, "n": len(ref), "ref": ref}, open(a.make_ref, "w"))
        print("wrote", a.make_ref, len(ref), "candidates")
        return
    ref = load_ref(a.ref, a.models, a.seed, len(jobs))
    report = compare(cases, ref)
    print("models %d  seed %d" % (a.models, a.seed))
    print_report(report)


def load_ref(path, models, seed, n_jobs):
    """Oracle results: the full file of --make-ref or the compact committed fixture (llk, status, rate x length)."""
    import gzip
    import json
    d = json.load(gzip.open(path, "rt") if path.endswith(".gz") else open(path))
    assert d["models"] == models and d["seed"] == seed and d["n"] == n_jobs, "reference file made with other settings"
    return [(r[0], None, r[1], r[2]) if len(r) == 3 else tuple(r) for r in d["ref"]]


def compare(cases, ref):
    """Evaluate every model as one batch through the C ABI and compare candidate by candidate."""
    from parity import llk_tol
    from misti_amd.engine import Engine
    stats = dict(candidates=0, both_fail=0, status_mismatch=0, regular=0, regular_within_tol=0, loose=0)
    worst_reg = worst_loose = 0.0
    bad = []
    loose = []
    pos = 0
    for ci, c in enumerate(cases):
        n = len(c["split"])
        with Engine(c["times"], c["lh"], c["bands"], c["pulses"], n_param=c["P"], sample_date=c["sd"], **c["flags"]) as e:
            r = e.evaluate(c["split"], c["params"], [c["sfs"]])
        # default fit with anything that mixes the pair states (a band or a pulse): the reference's bounded solver stops
        # on gtol far from the root and a 2^-48 input perturbation moves its llh by 1e-8..1e-4
        default_mig = (not c["flags"]["cpfit"]) and (not c["flags"]["true_eps"]) and (len(c["bands"]) > 0 or len(c["pulses"]) > 0)
        for k in range(n):
            o_llk, o_jafs, o_st, run = ref[pos + k]
            stats["candidates"] += 1
            noisy = run >= 5.0 or default_mig
            if o_st != 0 or r.status[k] != 0:
                if o_st == r.status[k] or (o_st != 0 and r.status[k] != 0):
                    stats["both_fail"] += 1
                elif not noisy:
                    stats["status_mismatch"] += 1
                    bad.append(("status", o_st, int(r.status[k]), float(c["split"][k]), c["flags"]))
                continue
            err = abs(r.llk[k, 0] - o_llk)
            rel = err / abs(o_llk)
            if not noisy:
                stats["regular"] += 1
                worst_reg = max(worst_reg, rel)
                if err <= llk_tol(o_llk, c["sfs"], o_jafs if o_jafs is not None else r.jafs[k], c["flags"]["unfolded"]):
                    stats["regular_within_tol"] += 1
                elif rel > 1e-7:
                    bad.append(("llk", rel, float(c["split"][k]), c["flags"], c["bands"], c["pulses"]))
            else:
                stats["loose"] += 1
                worst_loose = max(worst_loose, rel)
                loose.append((rel, ci, k, float(c["split"][k]), run, c["flags"]["cpfit"]))
        pos += n
    loose.sort(reverse=True)
    return dict(stats=stats, worst_regular=worst_reg, worst_loose=worst_loose, bad=bad, loose=loose)


def print_report(rep):
    stats, worst_reg, worst_loose, bad, loose = rep["stats"], rep["worst_regular"], rep["worst_loose"], rep["bad"], rep["loose"]
    print(stats)
    print("worst relative llk error: regular %.3g   noise-driven (runaway rate / default fit with a band or a pulse) %.3g" % (worst_reg, worst_loose))
    print("outliers beyond 1e-7 or status mismatches among regular candidates: %d" % len(bad))
    for b in bad[:20]:
        print("  ", b)
    print("largest differences in the noise-driven class (rel, model, candidate, split, rate x length, cpfit):")
    for b in loose[:12]:
        print("   %.3g  model %d cand %d split %.3f  rate x len %.3g  cpfit %s" % b)
    print("noise-driven class: %d above 1e-3, %d above 1e-6 of %d" % (sum(1 for b in loose if b[0] > 1e-3), sum(1 for b in loose if b[0] > 1e-6), len(loose)))


if __name__ == "__main__":
    main()
