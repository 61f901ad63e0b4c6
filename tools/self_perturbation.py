#!/usr/bin/env python3
"""The reference's own indeterminacy for the candidates of the random campaign (CPU only, no GPU).

ROUND 4: the campaign fixtures no longer use this tool's adaptive studies (4 perturbations for the noise class, deepened with --only where the
device disagreed: rounds 2-3) - tools/uniform_spread.py gives every candidate of the class the same 16 + 16 runs before the device is
consulted, and every first-pass outlier is run through /root/reference itself (tests/golden/make_golden.py --campaign-only).  This tool
stays for one-off studies with the ORACLE (`--only`, `--reference`, `--internal`); its side files <fixture>.spread.jsonl (git-ignored)
are what `tools/uniform_spread.py --validate` compares the baseline's spreads with.

For every candidate of `tools/random_campaign.py` that lies in the noise-driven class (corrected rate x
interval length >= 5, or default fit with a band or a pulse) the oracle - which reproduces the reference
bit for bit on the golden cases - is re-run on inputs perturbed by 2^-48 (`tests/parity.py: perturbed`,
kinds 0..K-1) and the largest relative change of its llh is recorded as that candidate's `spread`.
The parity contract (tests/parity.py) then allows the HIP path 10 x that spread for THAT candidate and
1e-9 everywhere else.

    python tools/self_perturbation.py --fixture tests/golden/campaign_seed1.json.gz --kinds 4 --procs 7
    python tools/self_perturbation.py --fixture ... --kinds 16 --only scratch/violators.json      # denser sampling for some
    python tools/self_perturbation.py --fixture ... --reference --only ...                        # the reference itself (here only)

    python tools/self_perturbation.py --fixture ... --internal 16 --only ...                      # one ulp in its own expm instead

Results are appended to <fixture>.spread.jsonl (resumable) and merged into the fixture with --merge:
every `ref` row becomes [llk, status, rate_x_len, spread, perturbed_runs_failed, kinds] for the candidates
that were studied, and [..., internal, internal_runs_failed, internal_runs] for those studied with --internal: the
largest relative change of the oracle's llh when its pair-chain matrix exponential (the reference's
scipy.linalg.expm at CorrectLambda.py:62) returns every entry moved by -1, 0 or +1 ulp at random
(the second measurement of tests/parity.py; tests/golden/internal_noise.py does the same to the reference itself).
"""
import argparse
import gzip
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

USE_REFERENCE = False


def eval_one(c, k, times, lh):
    """llh of candidate k of model c on the given grid (None = no value)."""
    par = list(c["params"][k]) if c["P"] else []
    if not USE_REFERENCE:
        from oracle.batch import oracle_eval
        llk, jafs, st, run = oracle_eval(times, lh, c["bands"], c["pulses"], c["flags"], c["sd"], float(c["split"][k]), par, [c["sfs"]])
        return None if st != 0 else float(llk[0])
    import contextlib
    import io
    import math
    import numpy
    numpy.mat = numpy.asmatrix
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    import MigrationInference as MI
    from oracle.batch import _mis_pus
    split = float(c["split"][k])
    s_int = int(math.floor(split))
    mis, pus = _mis_pus(c["bands"], c["pulses"], s_int + (1 if split != s_int else 0), par)
    order = [b[4] for b in c["bands"] if b[4] >= 0] + [b[3] for b in c["pulses"] if b[3] >= 0]
    f = c["flags"]
    with contextlib.redirect_stdout(io.StringIO()):
        m = MI.MigrationInference(list(times), [list(x) for x in lh], list(c["sfs"]), split, mis, pus, cpfit=f["cpfit"],
                                  trueEPS=f["true_eps"], smooth=f["smooth"], unfolded=f["unfolded"], sampleDate=c["sd"])
        llh = m.JAFSLikelihood([par[i] for i in order])
    return float(llh) if np.isfinite(llh) else None


def job(args):
    import parity
    idx, c, k, kinds, base = args
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if USE_REFERENCE:
            base = eval_one(c, k, c["times"], c["lh"])
        for kind in range(kinds):
            T, L = parity.perturbed(c["times"], c["lh"], kind)
            try:
                out.append(eval_one(c, k, T, L))
            except BaseException:                  # the reference exits on some structural errors
                out.append(None)
    return {"i": idx, "kinds": kinds, "base": base, "vals": out, "reference": USE_REFERENCE}


def internal_job(args):
    """`runs` evaluations of one candidate with one-ulp noise in the oracle's pair-chain expm."""
    import oracle.misti_oracle as om
    idx, c, k, runs, base = args
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for s in range(runs):
            rng = np.random.default_rng(7000 + s)

            def hook(e, rng=rng):
                kk = rng.integers(-1, 2, e.shape)
                return np.where(kk > 0, np.nextafter(e, np.inf), np.where(kk < 0, np.nextafter(e, -np.inf), e))
            om.EXPM_HOOK = hook
            try:
                out.append(eval_one(c, k, c["times"], c["lh"]))
            except BaseException:
                out.append(None)
            finally:
                om.EXPM_HOOK = None
    return {"i": idx, "internal_runs": runs, "base": base, "vals": out}


def noisy_class(c, ref_row):
    default_mig = (not c["flags"]["cpfit"]) and (not c["flags"]["true_eps"]) and (len(c["bands"]) > 0 or len(c["pulses"]) > 0)
    return ref_row[2] >= 5.0 or default_mig


def load_results(path):
    """Latest study per candidate: more kinds win; the reference's own runs are kept apart."""
    best, refruns = {}, {}
    if os.path.exists(path):
        for line in open(path):
            r = json.loads(line)
            tgt = refruns if r.get("reference") else best
            if r["i"] not in tgt or tgt[r["i"]]["kinds"] < r["kinds"]:
                tgt[r["i"]] = r
    return best, refruns


def summarise(r, base):
    fin = [v for v in r["vals"] if v is not None]
    if base is None:
        return None, len(r["vals"]) - len(fin), r["kinds"]        # a failing candidate: spread is meaningless; count the perturbed runs that have a value
    spread = max(abs(v - base) / abs(base) for v in fin) if fin else None
    return spread, len(r["vals"]) - len(fin), r["kinds"]


def main():
    global USE_REFERENCE
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", default=os.path.join(ROOT, "tests", "golden", "campaign_seed1.json.gz"))
    ap.add_argument("--kinds", type=int, default=4)
    ap.add_argument("--procs", type=int, default=7)
    ap.add_argument("--only", default="", help="JSON list of candidate indices to study (default: the whole noise-driven class)")
    ap.add_argument("--reference", action="store_true", help="run /root/reference itself instead of the oracle (build container only)")
    ap.add_argument("--internal", type=int, default=0, help="instead: this many runs with one-ulp noise in the oracle's pair-chain expm (needs --only)")
    ap.add_argument("--merge", action="store_true", help="fold <fixture>.spread.jsonl into the fixture and exit")
    a = ap.parse_args()
    USE_REFERENCE = a.reference
    import multiprocessing as mp
    import random_campaign as rc
    from threadpoolctl import threadpool_limits
    d = json.load(gzip.open(a.fixture, "rt"))
    rng = np.random.default_rng(d["seed"])
    cases = [rc.random_batch(rng) for _ in range(d["models"])]
    index = [(ci, k) for ci, c in enumerate(cases) for k in range(len(c["split"]))]
    assert len(index) == d["n"]
    side = a.fixture + ".spread.jsonl"
    have, refruns = load_results(side)
    side_int = a.fixture + ".internal.spread.jsonl"
    have_int = {}
    if os.path.exists(side_int):
        for line in open(side_int):
            r = json.loads(line)
            if r["i"] not in have_int or have_int[r["i"]]["internal_runs"] < r["internal_runs"]:
                have_int[r["i"]] = r
    if a.merge:
        n = 0
        for i, row in enumerate(d["ref"]):
            row = list(row[:3])
            if i in have:
                spread, nfail, kinds = summarise(have[i], row[0] if row[1] == 0 else None)
                row += [spread, nfail, kinds]
                n += 1
                if i in have_int and row[1] == 0:
                    r = have_int[i]
                    fin = [v for v in r["vals"] if v is not None]
                    row += [max(abs(v - row[0]) / abs(row[0]) for v in fin) if fin else None, len(r["vals"]) - len(fin), r["internal_runs"]]
            d["ref"][i] = row
        d["spread"] = ("rows of studied candidates: [llk, status, rate x length, spread, perturbed runs without a value, kinds"
                       " (, internal spread, internal runs without a value, internal runs)]; "
                       "tools/self_perturbation.py, perturbations of tests/parity.py")
        with gzip.open(a.fixture, "wt") as f:
            json.dump(d, f)
        print("merged %d studies into %s" % (n, a.fixture))
        return
    if a.only:
        todo = [int(i) for i in json.load(open(a.only))]
    else:
        todo = [i for i, (ci, k) in enumerate(index) if noisy_class(cases[ci], d["ref"][i])]
    if a.internal:
        assert a.only and not a.reference
        todo = [i for i in todo if d["ref"][i][1] == 0 and (i not in have_int or have_int[i]["internal_runs"] < a.internal)]
        print("%d candidates to study with %d noisy-expm runs each (oracle)" % (len(todo), a.internal), file=sys.stderr)
        jobs = [(i, cases[index[i][0]], index[i][1], a.internal, d["ref"][i][0]) for i in todo]
        with threadpool_limits(1), open(side_int, "a") as out:
            with mp.get_context("fork").Pool(a.procs) as pool:
                for n, r in enumerate(pool.imap_unordered(internal_job, jobs, chunksize=2)):
                    out.write(json.dumps(r) + "\n")
                    out.flush()
                    if n % 100 == 0:
                        print("studied %d / %d" % (n, len(jobs)), file=sys.stderr, flush=True)
        print("done: %s" % side_int)
        return
    tgt = refruns if a.reference else have
    todo = [i for i in todo if i not in tgt or tgt[i]["kinds"] < a.kinds]
    print("%d candidates to study with %d perturbations each (%s)" % (len(todo), a.kinds, "reference" if a.reference else "oracle"), file=sys.stderr)
    jobs = [(i, cases[index[i][0]], index[i][1], a.kinds, d["ref"][i][0] if d["ref"][i][1] == 0 else None) for i in todo]
    with threadpool_limits(1), open(side, "a") as out:
        with mp.get_context("fork").Pool(a.procs) as pool:
            for n, r in enumerate(pool.imap_unordered(job, jobs, chunksize=2)):
                out.write(json.dumps(r) + "\n")
                out.flush()
                if n % 200 == 0:
                    print("studied %d / %d" % (n, len(jobs)), file=sys.stderr, flush=True)
    print("done: %s" % side)


if __name__ == "__main__":
    main()
