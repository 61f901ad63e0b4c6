#!/usr/bin/env python3
"""The uniform indeterminacy protocol of round 4 for the random campaign (CPU only, no GPU, seconds per fixture).

Fixed BEFORE the device is consulted and never deepened afterwards: EVERY candidate of the noise class -
    corrected rate x interval length >= 5, or default fit with a band or a pulse, or "correction failed" in the oracle -
gets exactly K = 16 re-runs on inputs perturbed by 2^-48 (tests/parity.py: perturbed, kinds 0..15) and exactly K = 16 re-runs with one
ulp of noise in the pair chain's matrix exponential (generator seeds 7000..7015); `spread` / `internal` = the largest relative change
of the llh over those runs.  Candidates outside the class get no spread at all: clause 1 (1e-9) or OUTSIDE.

    python tools/uniform_spread.py --fixture tests/golden/campaign_seed5.json.gz            # writes the spreads into the fixture
    python tools/uniform_spread.py --fixture ... --validate                                  # against the oracle's own runs, where they exist

The values (llk, status) of a fixture are the NumPy/SciPy ORACLE's (tools/random_campaign.py --make-ref; bit for bit the reference on the
goldens).  The 32 re-runs per candidate are the COMPILED BASELINE's (oracle/cpu/misti_cpu.cpp: the reference's algorithm restated - Pade
expm, LU inverse, SciPy's TRF - pinned on the same goldens): the oracle needs ~2 core-seconds per run, i.e. ~45 core-hours per fixture at
this depth, the baseline a few core-seconds for the lot.  A spread is a statistic of the METHOD's rounding noise, not a value: --validate
compares, candidate by candidate, the baseline's spread over kinds 0..3 with the oracle's over the same four kinds (the round-3 studies,
<fixture>.spread.jsonl) and prints the distribution of the ratio."""
import argparse
import ctypes as C
import gzip
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

K = 16


def noisy_class(c, row, version=1):
    """version 1 (seeds 1-5): rate x length >= 5, default fit WITH a band or pulse, or a failed correction.  version 2 (seed 6 on): every
    default-fit candidate - round 4's reference runs showed that the bounded fit of an interval WITHOUT migration (CorrectLambda.py:253-264)
    is determined to 1e-9 ... 4e-9 only (nine of the first pass's twenty outliers)."""
    default_fit = (not c["flags"]["cpfit"]) and (not c["flags"]["true_eps"])
    default_mig = default_fit and (len(c["bands"]) > 0 or len(c["pulses"]) > 0)
    return row[2] >= 5.0 or (default_fit if version >= 2 else default_mig) or row[1] == 2


def baseline(c, sel, times, lh, threads):
    from oracle.cpu_baseline import cpu_eval
    par = None if not c["P"] else np.asarray(c["params"])[sel]
    llk, _, st, _, _ = cpu_eval(times, lh, c["bands"], c["pulses"], c["flags"], c["sd"], np.asarray(c["split"])[sel], par, [c["sfs"]], c["P"], threads=threads)
    return np.where(st == 0, llk[:, 0], np.nan)


def study(c, sel, threads, kinds=K, runs=K):
    """Per selected candidate of model c: (spread, runs without a value, kinds, internal spread, its runs without a value, runs, base has a value)."""
    import parity
    from oracle import cpu_baseline
    lib = cpu_baseline.load()
    lib.misti_cpu_set_expm_noise(C.c_int(-1))
    base = baseline(c, sel, c["times"], c["lh"], threads)
    pert = np.array([baseline(c, sel, *parity.perturbed(c["times"], c["lh"], k), threads) for k in range(kinds)])       # [kinds][n]
    inner = []
    for s in range(runs):
        lib.misti_cpu_set_expm_noise(C.c_int(7000 + s))
        inner.append(baseline(c, sel, c["times"], c["lh"], threads))
    lib.misti_cpu_set_expm_noise(C.c_int(-1))
    inner = np.array(inner).reshape(runs, len(sel))
    pert = pert.reshape(kinds, len(sel))
    out = []
    for j in range(len(sel)):
        b = base[j]
        def rel(vals):
            fin = vals[np.isfinite(vals)]
            if not np.isfinite(b) or len(fin) == 0:
                return None
            return float(np.max(np.abs(fin - b)) / abs(b))
        out.append((rel(pert[:, j]), int(np.isnan(pert[:, j]).sum()), kinds, rel(inner[:, j]), int(np.isnan(inner[:, j]).sum()), runs, bool(np.isfinite(b))))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", required=True)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--validate", action="store_true")
    ap.add_argument("--class-version", type=int, default=1, help="definition of the noise class (see noisy_class): 1 for seeds 1-5, 2 from seed 6 on")
    a = ap.parse_args()
    import random_campaign as rc
    d = json.load(gzip.open(a.fixture, "rt"))
    rng = np.random.default_rng(d["seed"])
    cases = [rc.random_batch(rng) for _ in range(d["models"])]
    pos = 0
    if a.validate:
        side = a.fixture + ".spread.jsonl"
        have = {}
        for line in open(side):
            r = json.loads(line)
            if not r.get("reference") and r["kinds"] >= 4 and (r["i"] not in have or have[r["i"]]["kinds"] < r["kinds"]):
                have[r["i"]] = r
        ratios, both_zero, flips_o, flips_b = [], 0, 0, 0
        for c in cases:
            n = len(c["split"])
            sel = [k for k in range(n) if (pos + k) in have and d["ref"][pos + k][1] == 0]
            if sel:
                res = study(c, np.array(sel), a.threads, kinds=4, runs=0)
                for k, r in zip(sel, res):
                    o = have[pos + k]
                    base = d["ref"][pos + k][0]
                    fin = [v for v in o["vals"][:4] if v is not None]
                    so = max(abs(v - base) / abs(base) for v in fin) if fin else None
                    flips_o += len(fin) < 4
                    flips_b += r[1] > 0
                    if so is None or r[0] is None:
                        continue
                    if so == 0 and r[0] == 0:
                        both_zero += 1
                    elif so > 0 and r[0] > 0:
                        ratios.append(r[0] / so)
            pos += n
        q = np.quantile(np.log10(ratios), [0.05, 0.1, 0.25, 0.5, 0.75, 0.9, 0.95])
        print("%s: baseline spread / oracle spread over the same four perturbations (kinds 0..3), %d candidates with both > 0 (%d with both = 0);"
              % (os.path.basename(a.fixture), len(ratios), both_zero))
        print("   ratio quantiles 5/10/25/50/75/90/95 %%: " + " ".join("%.2f" % 10 ** v for v in q))
        print("   candidates whose perturbed runs lose their value: oracle %d, baseline %d" % (flips_o, flips_b))
        return
    n_class = 0
    for c in cases:
        n = len(c["split"])
        rows = [list(d["ref"][pos + k][:3]) for k in range(n)]
        sel = [k for k in range(n) if noisy_class(c, rows[k], a.class_version)]
        if sel:
            for k, r in zip(sel, study(c, np.array(sel), a.threads)):
                spread, nfail, kinds, inner, ifail, runs, has = r
                if rows[k][1] == 0:
                    rows[k] += [spread, nfail, kinds, inner, ifail, runs]
                else:                       # a failure in the oracle: what counts is whether perturbed runs find a value (kinds - nfail of them)
                    rows[k] += [None, nfail, kinds]
                n_class += 1
        for k in range(n):
            d["ref"][pos + k] = rows[k]
        pos += n
    d["class_version"] = a.class_version
    d["protocol"] = ("uniform, round 4: every candidate with rate x length >= 5, " + ("default fit (with or without migration)" if a.class_version >= 2 else "default fit with a band or pulse") +
                     ", or 'correction failed' has exactly %d runs on inputs "
                     "perturbed by 2^-48 (kinds 0..%d) and %d runs with one ulp of noise in the pair chain's expm, by the compiled baseline (oracle/cpu/misti_cpu.cpp); "
                     "fixed before the device was consulted (tools/uniform_spread.py).  llk / status: the NumPy/SciPy oracle" % (K, K - 1, K))
    d["spread"] = "rows of the noise class: [llk, status, rate x length, spread, perturbed runs without a value, kinds, internal spread, internal runs without a value, internal runs]"
    with gzip.open(a.fixture, "wt") as f:
        json.dump(d, f)
    print("%s: %d of %d candidates in the noise class, %d + %d runs each" % (a.fixture, n_class, d["n"], K, K))


if __name__ == "__main__":
    main()
