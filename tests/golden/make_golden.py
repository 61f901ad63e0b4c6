#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE (``/root/reference``).

Runs only in the build container (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

and writes ``tests/golden/golden_*.json``: plain data - inputs of the hot path
(interval lengths, PSMC rates, JSFS row, split time, -mi/-pu descriptors, flags,
parameter vector) and what the reference's ``MigrationInference`` returned for
them (llh, llh_const, expected JAFS, corrected rates ``lc``, pair-state trace
``Pr``).  Also dumps the reference reader's ``InputData`` for synthetic PSMC
files to pin ``misti_amd.io.read_psmc``.

Reference environment: Python 3.10.12, NumPy 2.2.6 (``numpy.mat`` alias
restored), SciPy 1.15.3.  The inputs are synthetic (``misti_amd.synth``).
"""
import contextlib
import io
import json
import os
import random
import sys
import tempfile
import time

import numpy

numpy.mat = numpy.asmatrix            # NumPy >= 2 removed the alias the reference imports
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import migrationIO                     # noqa: E402  (reference)
import MigrationInference as MI        # noqa: E402  (reference)
import CorrectLambda as CL              # noqa: E402  (reference)
from scipy import optimize             # noqa: E402
from misti_amd import synth            # noqa: E402
import parity                          # noqa: E402  (tests/parity.py: the perturbation family and the contract's constants)


def run_reference(times, lambdas, sfs, split, mi, pu, kw, params):
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        m = MI.MigrationInference(list(times), [list(x) for x in lambdas], list(sfs), split,
                                  [list(x) for x in mi], [list(x) for x in pu], **kw)
        llh = m.JAFSLikelihood(list(params))
    ok = bool(numpy.isfinite(llh))
    rec = {
        "llh": float(llh) if ok else None,
        "llh_const": float(m.llh_const),
        "numT": m.numT, "splitT": m.splitT,
        "stdout": out.getvalue().strip().splitlines()[-1:] if not ok else [],
    }
    if ok:
        rec["JAFS"] = [float(v) for v in m.JAFS]
        rec["lc"] = [[float(a), float(b)] for a, b in m.lc]
        rec["Pr"] = [[[float(c) for c in r] for r in p] for p in m.Pr]
    return rec


PERTURB = parity.PERTURB      # relative input perturbation used to measure the reference's own conditioning


def perturbation_study(times, lambdas, sfs, split, mi, pu, kw, params, llh, kinds):
    """The reference's own indeterminacy for this case: it is re-run on inputs perturbed by 2^-48
    (relative; `parity.perturbed`, kinds 0..kinds-1).  The lambda-correction solves stop on SciPy's
    tests after a few finite-difference trust-region steps; where the residual is flat (the pair has
    all but coalesced inside the interval) the stopping point is decided by rounding noise and the
    reference's own output is not determined to 1e-9.  Returns
      sens       max over kinds 0-2 of |llh' - llh| / |llh| / 2^-48 (None where one of them fails) - round 1's figure
      spread     max over all kinds run of |llh' - llh| / |llh| (finite perturbed runs)
      pert_fail  number of perturbed runs that ended in "correction failed"
      pert_llh   the perturbed values themselves (None = failed)."""
    vals = []
    for kind in range(kinds):
        T, L = parity.perturbed(times, lambdas, kind)
        vals.append(run_reference(T, L, sfs, split, mi, pu, kw, params)["llh"])
    base = [v for v in vals[:3]]
    sens = None if any(v is None for v in base) else max(abs(v - llh) / abs(llh) / PERTURB for v in base)
    fin = [v for v in vals if v is not None]
    spread = max(abs(v - llh) / abs(llh) for v in fin) if fin else None
    return sens, spread, sum(1 for v in vals if v is None), vals


def case(name, times, lambdas, sfs, split, mi=(), pu=(), params=(), **kw):
    t0 = time.time()
    rec = run_reference(times, lambdas, sfs, split, mi, pu, kw, params)
    if name != "tmp":
        if rec["llh"] is not None:
            sens, spread, nfail, vals = perturbation_study(times, lambdas, sfs, split, mi, pu, kw, params, rec["llh"], parity.N_KINDS_BASE)
            if sens is None or sens >= parity.SENS_DETERMINED:      # reference-indeterminate: sample its spread more densely
                _, spread, nfail, vals = perturbation_study(times, lambdas, sfs, split, mi, pu, kw, params, rec["llh"], parity.N_KINDS_DEEP)
            rec["sens"], rec["spread"], rec["pert_fail"], rec["pert_llh"] = sens, spread, nfail, vals
        else:
            # a failing case: does the reference flip to a value under the same perturbations?
            vals = [run_reference(*parity.perturbed(times, lambdas, k), sfs, split, mi, pu, kw, params)["llh"] for k in range(parity.N_KINDS_BASE)]
            rec["pert_finite"] = sum(1 for v in vals if v is not None)
    return {"name": name,
            "in": {"times": list(times), "lambdas": [list(x) for x in lambdas], "sfs": list(sfs),
                   "split": split, "mi": [list(x) for x in mi], "pu": [list(x) for x in pu],
                   "kw": kw, "params": list(params)},
            "out": rec, "ref_seconds": round(time.time() - t0, 3)}


# ---- solver traces ---------------------------------------------------------------------------
class SolverTrace:
    """Log every scipy.optimize.least_squares call the reference makes (CorrectLambda.py:85,260,303,305):
    which interval, the start, the result, nfev, status and the TRIAL POINTS - the evaluation points that
    are not finite-difference points of the 2-point Jacobian (after each accepted point SciPy evaluates
    x + h_i e_i for i = 0..n-1, h_i = sqrt(eps) max(1, |x_i|) sign(x_i), flipped at a bound)."""

    SITES = {"LambdaSystem": "two_pop_ect", "LambdaSystem1": "two_pop_cp", "LambdaSystemNoMigration": "no_migration", "<lambda>": "single_pop"}

    def __enter__(self):
        self.solves = []
        self.seq = -1
        self.orig_ls = optimize.least_squares
        self.orig_si = CL.CorrectLambda.SetInterval
        tracer = self

        def set_interval(obj, lh, T, P0):
            tracer.seq += 1
            return tracer.orig_si(obj, lh, T, P0)

        def least_squares(fun, x0, *a, **kw):
            calls = []

            def logged(x, *aa, **kk):
                calls.append([float(v) for v in numpy.atleast_1d(x)])
                return fun(x, *aa, **kk)
            res = tracer.orig_ls(logged, x0, *a, **kw)
            n = len(calls[0])
            trials, i = [], 0
            while i < len(calls):
                xt = calls[i]
                trials.append(xt)
                fd = calls[i + 1:i + 1 + n]
                is_fd = len(fd) == n
                for j, xf in enumerate(fd):
                    h = 1.4901161193847656e-08 * max(1.0, abs(xt[j]))
                    for k in range(n):
                        d = abs(xf[k] - xt[k])
                        if (k == j and not (0.5 * h <= d <= 2.0 * h)) or (k != j and d != 0.0):
                            is_fd = False
                i += 1 + (n if is_fd else 0)
            assert len(trials) == res.nfev, (len(trials), res.nfev)
            tracer.solves.append({"seq": tracer.seq, "site": tracer.SITES.get(getattr(fun, "__name__", "?"), "?"),
                                  "x0": [float(v) for v in numpy.atleast_1d(x0)], "x": [float(v) for v in res.x],
                                  "nfev": int(res.nfev), "status": int(res.status), "optimality": float(res.optimality),
                                  "trials": trials})
            return res
        optimize.least_squares = least_squares
        CL.CorrectLambda.SetInterval = set_interval
        return self

    def __exit__(self, *exc):
        optimize.least_squares = self.orig_ls
        CL.CorrectLambda.SetInterval = self.orig_si


def traced(c):
    """Re-run golden case `c` under SolverTrace; the traced run must reproduce the recorded llh bit for bit."""
    i = c["in"]
    with SolverTrace() as tr:
        rec = run_reference(i["times"], i["lambdas"], i["sfs"], i["split"], i["mi"], i["pu"], i["kw"], i["params"])
    assert rec["llh"] == c["out"]["llh"], (c["name"], rec["llh"], c["out"]["llh"])
    # interval index of every solve: the two-population intervals call SetInterval once each (t = seq); after the
    # split only the default fit does, skipping zero-length intervals (MigrationInference.py:355-362)
    splitT, numT = rec["splitT"], rec["numT"]
    times = list(i["times"])
    frac = i["split"] % 1
    if frac != 0.0:                                   # the grid the reference works on (:89-99)
        s0 = int(i["split"])
        t1 = frac * times[s0]
        times[s0:s0 + 1] = [t1, times[s0] - t1]
    post = [t for t in range(splitT, numT - 1) if times[t] != 0]
    n_two = splitT if not i["kw"].get("trueEPS") else 0
    for sv in tr.solves:
        sv["t"] = sv["seq"] if sv["seq"] < n_two else post[sv["seq"] - n_two]
    return {"name": c["name"], "numT": numT, "splitT": splitT, "solves": tr.solves}


def wants_trace(c):
    """Every reference-indeterminate golden, every default-fit case with a band or a pulse, and a few determined controls."""
    o, kw, i = c["out"], c["in"]["kw"], c["in"]
    if o["llh"] is None:
        return False
    indet = o.get("sens") is None or o["sens"] >= parity.SENS_DETERMINED
    default_mig = not kw.get("cpfit") and not kw.get("trueEPS") and (len(i["mi"]) > 0 or len(i["pu"]) > 0)
    return indet or default_mig or c["name"] in ("A1", "A3", "A7", "B6", "c1_n32_default", "c2_n128_st60_r0.2", "c4_n128_default_st44.5")


def anchors():
    """SURVEY.md appendix A."""
    T = [0.01, 0.02, 0.04, 0.08, 0.16, 0.32, 0.64]
    L = [[1, 2], [1, 2], [0.8, 1.5], [0.8, 1.5], [1.2, 1.0], [1.2, 1.0], [0.9, 0.9], [0.7, 0.7]]
    S = [100000, 900, 250, 1000, 600, 400, 260, 410]
    two = [[1, 1, 5, 0.3, 1], [2, 1, 5, 0.1, 1]]
    c = []
    c.append(case("A1", T, L, S, 5, smooth=True))
    c.append(case("A2", T, L, S, 5, smooth=True, cpfit=True))
    c.append(case("A3", T, L, S, 5, two, params=[0.3, 0.1], smooth=True, cpfit=True, unfolded=True))
    c.append(case("A4", T, L, S, 5, two, params=[0.3, 0.1], smooth=True))
    c.append(case("A5", T, L, S, 5, [[1, 0, 5, 0.5, 0]], [[2, 3, 0.2, 0]], trueEPS=True, unfolded=True))
    c.append(case("A6", T, L, S, 5, [[1, 2, 5, 0.3, 0]], smooth=True, cpfit=True, sampleDate=2))
    c.append(case("A7", T, L, S, 4.5, smooth=True, cpfit=True))
    c.append(case("A8", T, L, S, 5, [[1, 1, 5, -0.1, 1]], params=[-0.1], smooth=True, cpfit=True))
    c.append(case("A9", T, L, S, 5, [[2, 0, 5, 0.2, 0]], [[1, 2, 0.1, 1]], params=[0.1], cpfit=True))
    # extra small cases: every flag combination on the literal inputs
    c.append(case("B1", T, L, S, 5, two, params=[0.05, 0.7], cpfit=True))
    c.append(case("B2", T, L, S, 5, two, params=[0.05, 0.7], trueEPS=True, cpfit=True, smooth=True))
    c.append(case("B3", T, L, S, 5, two, params=[0.05, 0.7], trueEPS=True))
    c.append(case("B4", T, L, S, 3, [[1, 0, 3, 1.5, 1]], [[1, 1, 0.3, 1]], params=[1.5, 0.3], cpfit=True, smooth=True))
    c.append(case("B5", T, L, S, 6, [[2, 1, 4, 0.2, 0]], [[2, 4, 0.5, 0]], cpfit=True, smooth=True, unfolded=True))
    c.append(case("B6", T, L, S, 5.25, [[1, 0, 6, 0.4, 1]], params=[0.4], cpfit=True, smooth=True))
    c.append(case("B7", T, L, S, 5, two, params=[0.3, 0.1], smooth=True, cpfit=True, mixtureTH=0.9))
    c.append(case("B8", T, L, S, 5, [[1, 0, 5, 0.5, 0]], [[1, 0, 1.0, 0]], cpfit=True, smooth=True, sampleDate=0))
    c.append(case("B9", T, L, S, 5, [[1, 3, 5, 0.5, 0]], [], cpfit=True, smooth=True, sampleDate=3))
    c.append(case("B10", T, L, S, 5, [], [[2, 5, 0.5, 0]], cpfit=True, smooth=True, sampleDate=5))
    c.append(case("B11", T, L, S, 7, [[1, 0, 7, 0.2, 0], [2, 0, 7, 0.1, 0]], cpfit=True, smooth=True))
    c.append(case("B12", T, L, S, 1, cpfit=True, smooth=True))
    c.append(case("B13", T, L, S, 0, cpfit=True, smooth=True))
    c.append(case("B14", T, L, S, 5, two, params=[30.0, 0.1], smooth=True, cpfit=True))
    # mixture threshold (-mth) that does not fire / fires late: the guard of SolveLambdaSystem (CorrectLambda.py:267-272)
    c.append(case("B15", T, L, S, 5, two, params=[0.3, 0.1], smooth=True, cpfit=True, mixtureTH=0.05))
    c.append(case("B16", T, L, S, 5, two, params=[2.0, 1.5], smooth=True, cpfit=True, mixtureTH=0.3))
    c.append(case("B17", T, L, S, 5, two, params=[0.3, 0.1], smooth=True, mixtureTH=0.05))
    c.append(case("B18", T, L, S, 5, two, params=[0.3, 0.1], smooth=True, cpfit=True, mixtureTH=1.5))       # fires: correction failed
    c.append(case("B19", T, L, S, 5, two, params=[3.0, 3.0], smooth=True, cpfit=True, mixtureTH=0.5))       # fires after the pairs have mixed
    return c


def reader_dumps(tmp):
    """InputData as the reference's own ReadPSMC produces it, for synthetic PSMC text."""
    out = []
    for n1, n2, sdate, hl in [(16, 17, 0.0, None), (64, 65, 0.0, None), (64, 64, 40000.0, (0.0, 0.1)),
                              (24, 31, 12000.0, (0.05, 0.2))]:
        t1 = synth.psmc_text(n1, 1, synth.THETA_1, rounds=2)
        t2 = synth.psmc_text(n2, 2, synth.THETA_2, rounds=2)
        f1, f2 = os.path.join(tmp, "g1.psmc"), os.path.join(tmp, "g2.psmc")
        open(f1, "w").write(t1)
        open(f2, "w").write(t2)
        migrationIO.Units.hetloss1, migrationIO.Units.hetloss2 = (hl or (0.0, 0.0))
        d = migrationIO.ReadPSMC(f1, f2, sdate)
        migrationIO.Units.hetloss1, migrationIO.Units.hetloss2 = 0.0, 0.0
        out.append({"n1": n1, "n2": n2, "sdate": sdate, "hetloss": list(hl) if hl else None,
                    "psmc1": t1, "psmc2": t2,
                    "times": d.times, "lambdas": d.lambdas, "scaleTime": d.scaleTime, "theta": d.theta,
                    "rho": d.rho, "sampleDateDiscr": d.sampleDateDiscr, "Tpsmc": d.Tpsmc})
    return out


def synthetic_cases():
    """numT = 32 and numT = 128 cases shaped like BASELINE.json's configs."""
    c = []
    # ---- config 1: numT=32, split 20, no migration --------------------------
    inp32 = synth.psmc_pair(16, 17)
    t32, lh32, _ = synth.self_consistent(inp32, 20)
    truth = case("tmp", t32, lh32, [1] * 8, 20, trueEPS=True, cpfit=True, unfolded=True)
    sfs32 = synth.counts_from_spectrum(truth["out"]["JAFS"])
    for nm, kw in [("default", dict(smooth=True)), ("cpfit", dict(smooth=True, cpfit=True)),
                   ("cpfit_uf_nosmooth", dict(cpfit=True, unfolded=True)),
                   ("trueEPS_cpfit", dict(trueEPS=True, cpfit=True, smooth=True))]:
        c.append(case("c1_n32_" + nm, t32, lh32, sfs32, 20, **kw))
    # raw random PSMC rates (not self-consistent): correction stress
    c.append(case("c1_n32_rawpsmc_cpfit", inp32.times, inp32.lambdas, sfs32, 20, smooth=True, cpfit=True))
    c.append(case("c1_n32_rawpsmc_default", inp32.times, inp32.lambdas, sfs32, 20, smooth=True))

    # ---- config 2: numT=128, one band -mi 1 4 {st} {r} 1, --cpfit ------------
    inp = synth.psmc_pair(64, 65)
    true_split, true_rate = 64, 0.2
    t128, lh128, _ = synth.self_consistent(inp, true_split, [[1, 4, true_split, true_rate, 0]])
    truth = case("tmp", t128, lh128, [1] * 8, true_split, [[1, 4, true_split, true_rate, 0]],
                 trueEPS=True, cpfit=True, unfolded=True)
    sfs128 = synth.counts_from_spectrum(truth["out"]["JAFS"])
    rng = random.Random(5)
    for st in (30, 48, 60, 64, 70, 93):
        for r in (1e-3, 0.0316, 0.2, 1.0):
            c.append(case("c2_n128_st%d_r%g" % (st, r), t128, lh128, sfs128, st,
                          [[1, 4, st, r, 1]], params=[r], smooth=True, cpfit=True))
    c.append(case("c2_n128_uf", t128, lh128, sfs128, 64, [[1, 4, 64, 0.2, 1]], params=[0.2],
                  smooth=True, cpfit=True, unfolded=True))
    c.append(case("c2_n128_pop2", t128, lh128, sfs128, 64, [[2, 4, 64, 0.2, 1]], params=[0.37],
                  smooth=True, cpfit=True))
    # ---- config 3: two optimised bands ------------------------------------
    for k in range(6):
        p = [10 ** rng.uniform(-3, 0), 10 ** rng.uniform(-3, 0)]
        c.append(case("c3_n128_two_bands_%d" % k, t128, lh128, sfs128, 64,
                      [[1, 4, 64, 0.1, 1], [2, 10, 64, 0.1, 1]], params=p, smooth=True, cpfit=True))
    c.append(case("c3_n128_two_bands_default", t128, lh128, sfs128, 64,
                  [[1, 4, 64, 0.1, 1], [2, 10, 64, 0.1, 1]], params=[0.2, 0.05], smooth=True))
    c.append(case("c3_n128_two_bands_trueEPS", t128, lh128, sfs128, 64,
                  [[1, 4, 64, 0.1, 1], [2, 10, 64, 0.1, 1]], params=[0.2, 0.05], smooth=True, trueEPS=True))
    # ---- config 4: no migration, split scan incl. fractional, replicates ---
    t4, lh4, _ = synth.self_consistent(inp, 50)
    truth = case("tmp", t4, lh4, [1] * 8, 50, trueEPS=True, cpfit=True, unfolded=True)
    row = synth.counts_from_spectrum(truth["out"]["JAFS"])
    chunks = synth.chunk_rows(row, 20)
    from misti_amd import io as mio
    table = mio.bootstrap_table(chunks, 4, random.Random(3))
    for st in (20, 44.5, 50, 50.25, 77):
        for b in (0, 2):
            c.append(case("c4_n128_st%g_bs%d" % (st, b), t4, lh4, table[b], st, smooth=True, cpfit=True))
    c.append(case("c4_n128_default_st50", t4, lh4, table[0], 50, smooth=True))
    c.append(case("c4_n128_default_st44.5", t4, lh4, table[1], 44.5, smooth=True))
    # ---- config 5: ancient second genome + hetloss, band + pulse ----------
    inp5 = synth.psmc_pair(64, 64, sample_date=40000.0, units=mio.Units(hetloss2=0.1))
    sd = inp5.sampleDateDiscr
    t5, lh5, _ = synth.self_consistent(inp5, 80, [[1, sd + 2, 80, 0.15, 0]], [[2, sd + 10, 0.1, 0]])
    truth = case("tmp", t5, lh5, [1] * 8, 80, [[1, sd + 2, 80, 0.15, 0]], [[2, sd + 10, 0.1, 0]],
                 trueEPS=True, cpfit=True, unfolded=True, sampleDate=sd)
    sfs5 = synth.counts_from_spectrum(truth["out"]["JAFS"])
    for st in (70, 80, 95):
        for r, q in ((0.05, 0.02), (0.15, 0.1), (0.6, 0.4)):
            c.append(case("c5_n128_sd%d_st%d_r%g_q%g" % (sd, st, r, q), t5, lh5, sfs5, st,
                          [[1, sd + 2, st, r, 1]], [[2, sd + 10, q, 1]], params=[r, q],
                          smooth=True, cpfit=True, sampleDate=sd))
    c.append(case("c5_n128_default", t5, lh5, sfs5, 80, [[1, sd + 2, 80, 0.15, 1]], [[2, sd + 10, 0.1, 1]],
                  params=[0.15, 0.1], smooth=True, sampleDate=sd))
    return c


MS_STRINGS = [
    # SURVEY appendix A anchor (TestModel route)
    ("ms_anchor", True, "4 100 -t 15000 -r 1920 30000000 -l -I 2 2 2 -n 1 10 -n 2 4.5 -em 0.0 1 2 2.0 -em 0.0 2 1 3.0 "
                        "-eN 0.025 0.2 -ej 0.045 2 1 -eN 0.175 3 -eN 0.625 1.8 -eN 3 3.2 -eN 8 5.5"),
    # band chaining (two -em of one population), per-population size changes, folded
    ("ms_chain", False, "4 1 -t 100 -I 2 2 2 -n 1 1.5 -n 2 0.7 -em 0.0 1 2 1.0 -em 0.02 1 2 0.25 -em 0.01 2 1 0.5 "
                        "-en 0.03 1 2.0 -en 0.05 2 0.4 -ej 0.11 2 1 -eN 0.4 2.5 -eN 1.5 1.0"),
    # a pulse (-es), migration ending before the split (-em ... 0), population 1 merged into 2
    ("ms_pulse", True, "4 1 -t 100 -I 2 2 2 -n 2 3.0 -em 0.0 2 1 1.5 -em 0.04 2 1 0 -es 0.02 1 0.8 -en 0.06 1 0.5 "
                       "-ej 0.2 1 2 -eN 0.5 2.0 -eN 2.0 4.0"),
    # no migration at all
    ("ms_nomig", True, "4 1 -t 100 -I 2 2 2 -n 1 2.0 -n 2 0.5 -en 0.05 1 1.0 -ej 0.3 2 1 -eN 0.6 1.7 -eN 3.0 0.8"),
]


def ms_cases():
    """ReadMS -> MigrationInference(trueEPS) -> JAFSLikelihood([]) -> CoalescentRates(): the TestModel.py route."""
    out = []
    for name, unfolded, text in MS_STRINGS:
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
            d = migrationIO.ReadMS(text)
            rec = {"name": name, "ms": text, "unfolded": unfolded,
                   "times": [float(v) for v in d.times], "lambdas": [[float(a), float(b)] for a, b in d.lambdas],
                   "divergenceTime": d.divergenceTime, "mi": [list(map(float, m)) for m in d.mi], "pu": [list(map(float, q)) for q in d.pu]}
            m = MI.MigrationInference(list(d.times), [list(x) for x in d.lambdas], [1] * 8, d.divergenceTime,
                                      [list(x) for x in d.mi], [list(x) for x in d.pu], unfolded=unfolded, trueEPS=True)
            llh = m.JAFSLikelihood([])
            rec["llh"] = float(llh)
            rec["JAFS"] = [float(v) for v in m.JAFS]
            rec["lc"] = [[float(a), float(b)] for a, b in m.lc]
            m.CoalescentRates()
            rec["forward_lh"] = [[float(a), float(b)] for a, b in m.lh]
            rec["forward_Pr"] = [[[float(c) for c in r] for r in p] for p in m.Pr]
        out.append(rec)
    return out


def sweep_cases():
    """The reference's own recommended sweep (README.md:110-115): four fixed bands whose rates change at interval
    {mc}, `-mi 1 0 {mc} a 0 -mi 2 0 {mc} b 0 -mi 1 {mc} {st} c 0 -mi 2 {mc} {st} d 0 -uf`, over st x mc x rates,
    one reference run per grid point.  The HIP path evaluates the whole grid in ONE call with per-candidate band
    bounds (misti_eval_batch's band_bounds).  numT = 32; the README's default fit and --cpfit."""
    inp = synth.psmc_pair(16, 17)
    st0, mc0 = 20, 9
    true_mi = [[1, 0, mc0, 0.3, 0], [2, 0, mc0, 0.1, 0], [1, mc0, st0, 0.05, 0], [2, mc0, st0, 0.4, 0]]
    t32, lh32, _ = synth.self_consistent(inp, st0, true_mi)
    truth = case("tmp", t32, lh32, [1] * 8, st0, true_mi, trueEPS=True, cpfit=True, unfolded=True)
    sfs = synth.counts_from_spectrum(truth["out"]["JAFS"])
    out = []
    for cp in (True, False):
        for st in (19, 20, 21, 22.5):
            end = int(st) + (1 if st % 1 else 0)
            for mc in (8, 9, 10, 11):
                for ri, rates in enumerate(([0.3, 0.1, 0.05, 0.4], [0.02, 0.5, 0.25, 0.0])):
                    mi = [[1, 0, mc, rates[0], 0], [2, 0, mc, rates[1], 0], [1, mc, end, rates[2], 0], [2, mc, end, rates[3], 0]]
                    kw = dict(smooth=True, unfolded=True)
                    if cp:
                        kw["cpfit"] = True
                    c = case("sw_%s_st%g_mc%d_r%d" % ("cp" if cp else "df", st, mc, ri), t32, lh32, sfs, st, mi, **kw)
                    c["sweep"] = {"st": st, "mc": mc, "rates": rates}
                    out.append(c)
    return out


# the candidates of the random campaign (tools/random_campaign.py, seed 1, 600 models) on which the HIP path stands
# worst against the ORACLE: every one outside the contract there, and the two largest factors inside it
CAMPAIGN_PICKS = ((1, 148, 12), (1, 515, 0), (1, 515, 2), (1, 515, 4), (1, 515, 6), (1, 583, 3), (1, 547, 2), (1, 202, 0), (1, 353, 0), (1, 198, 1),
                  (2, 35, 1), (2, 35, 9), (2, 35, 11), (2, 35, 21),       # (campaign seed, model, candidate)
                  # round 4: EVERY candidate the first pass of the uniform protocol (tools/uniform_spread.py: 16 + 16 runs of the compiled baseline
                  # for every candidate of the noise class, fixed before the device was consulted; profiles/r04_random_campaign_seed*.txt) left
                  # outside the contract - seed 5 is the new held-out fixture
                  (5, 282, 14), (5, 38, 7), (5, 38, 6), (5, 38, 5), (5, 545, 5), (5, 461, 0),
                  (1, 446, 1), (1, 446, 2), (1, 446, 3), (1, 446, 4), (1, 446, 5),
                  (2, 35, 0), (2, 35, 2), (2, 35, 8), (2, 35, 12), (2, 35, 14), (2, 35, 16),
                  (3, 234, 1), (4, 581, 14), (4, 307, 8),
                  # round 5: the contract's factor went from 10 to 3 (tests/parity.py: SELF_FACTOR) and the stiff two-way pair exponential became a
                  # closed form; EVERY candidate the first pass then leaves outside that was not already here (27; seed 6 is round 4's second held-out fixture)
                  (1, 229, 5), (1, 229, 7), (1, 229, 9), (1, 229, 11), (1, 229, 13), (1, 206, 21), (2, 16, 3), (2, 94, 1), (2, 35, 20), (2, 42, 0), (2, 215, 1),
                  (3, 417, 5), (3, 417, 6), (3, 140, 1), (3, 140, 5), (3, 140, 7), (3, 140, 11), (3, 140, 13), (3, 140, 15), (3, 140, 17),
                  (4, 303, 2), (4, 303, 3), (4, 303, 4), (4, 303, 6), (4, 303, 7), (5, 397, 1), (6, 313, 4),
                  # seed 7: the fixture generated AFTER all of round 5's changes and studies (held out); every candidate its first pass leaves outside (9)
                  (7, 421, 5), (7, 93, 0), (7, 93, 2), (7, 93, 4), (7, 93, 6), (7, 93, 8), (7, 93, 10), (7, 93, 12), (7, 312, 0),
                  # seed 8: a second fixture generated after everything (the round's last build); both candidates its first pass leaves outside
                  (8, 467, 2), (8, 589, 0))


def campaign_cases(kinds=64, have=None):
    """Those candidates run through the REFERENCE itself, with a denser perturbation study (`kinds` perturbed runs each)
    and solver traces: what the campaign measures against the oracle, measured against the reference.  `have`: cases of an
    earlier call by (seed, model, candidate) - kept as they are (--campaign-only --keep: the file grows by the new picks)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from random_campaign import random_batch
    models = {}
    for seed in sorted({p[0] for p in CAMPAIGN_PICKS}):
        rng = numpy.random.default_rng(seed)
        models[seed] = [random_batch(rng) for _ in range(600)]
    out = []
    for seed, mi_, k in CAMPAIGN_PICKS:
        if have and (seed, mi_, k) in have:
            out.append(have[(seed, mi_, k)])
            continue
        c = models[seed][mi_]
        st = float(c["split"][k])
        end = int(st) + (1 if st % 1 else 0)
        par = [float(v) for v in c["params"][k]] if c["P"] else []
        mi = [[pop + 1, start, end if e < 0 else e, par[P] if P >= 0 else v, 1 if P >= 0 else 0] for (pop, start, e, v, P) in c["bands"]]
        pu = [[pop + 1, t, par[P] if P >= 0 else v, 1 if P >= 0 else 0] for (pop, t, v, P) in c["pulses"]]
        f = c["flags"]
        kw = dict(smooth=f["smooth"], unfolded=f["unfolded"])
        if f["cpfit"]:
            kw["cpfit"] = True
        if f["true_eps"]:
            kw["trueEPS"] = True
        if c["sd"]:
            kw["sampleDate"] = c["sd"]
        split = int(st) if st % 1 == 0 else st
        g = case(("camp_m%d_c%d" if seed == 1 else "camp_s%d_m%%d_c%%d" % seed) % (mi_, k), c["times"], c["lh"], c["sfs"], split, mi, pu, par, **kw)
        o = g["out"]
        if o["llh"] is not None:
            _, o["spread"], o["pert_fail"], o["pert_llh"] = perturbation_study(c["times"], c["lh"], c["sfs"], split, mi, pu, kw, par, o["llh"], kinds)
            # determined or not is judged on ALL the perturbed runs here (three kinds miss a heavy tail: m547 moves by 3e-11 under
            # kinds 0-2 and by 1.9e-10 under kind 17)
            o["sens"] = None if o["pert_fail"] else o["spread"] / PERTURB
        o.pop("Pr", None)
        g["campaign"] = {"seed": seed, "model": mi_, "cand": k}
        out.append(g)
    return out


def raw(v):
    """JSON-able copy that keeps Python ints as ints and turns NumPy scalars into Python floats."""
    if isinstance(v, (list, tuple)) or hasattr(v, "tolist") and getattr(v, "ndim", 0) > 0:
        return [raw(x) for x in v]
    if isinstance(v, (bool, int)):
        return v
    return float(v)


def host_fixtures(tmp):
    """Reference outputs that pin the host-side pieces of the path (f2): the `#MiSTI2 ver 0.4` writer
    (migrationIO.OutputMigration :346-375), the bootstrap resampler under a fixed seed (BootstrapJAFS :506-524 as
    utils/generateJSFS_bs.py:39-48 drives it) and MiSTI.py's machine-read result line (:240) with its -o file."""
    import subprocess
    import types
    T = [0.01, 0.02, 0.04, 0.08, 0.16, 0.32, 0.64]
    L = [[1, 2], [1, 2], [0.8, 1.5], [0.8, 1.5], [1.2, 1.0], [1.2, 1.0], [0.9, 0.9], [0.7, 0.7]]
    S = [100000, 900, 250, 1000, 600, 400, 260, 410]
    two = [[1, 1, 5, 0.3, 1], [2, 1, 5, 0.1, 1]]
    writer = []
    for name, split, mi, pu, params, kw in [("A3", 5, two, [], [0.3, 0.1], dict(smooth=True, cpfit=True, unfolded=True)),
                                             ("A6", 5, [[1, 2, 5, 0.3, 0]], [], [], dict(smooth=True, cpfit=True, sampleDate=2)),
                                             ("A7", 4.5, [], [], [], dict(smooth=True, cpfit=True))]:
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink):
            m = MI.MigrationInference(list(T), [list(x) for x in L], list(S), split, [list(x) for x in mi], [list(x) for x in pu],
                                      thrh=[0.0521, 0.0104], **kw)
            m.JAFSLikelihood(list(params))
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            migrationIO.OutputMigration("", list(params), m, 1234.5, 0.75)
        text = out.getvalue()
        text = text[text.index("#MiSTI2"):]
        # ... and what the reference's own reader of that format makes of it (migrationIO.ReadMigration :377-504, no plotting)
        fm = os.path.join(tmp, "w_%s.mi" % name)
        open(fm, "w").write(text)
        with contextlib.redirect_stdout(io.StringIO()):
            rd = migrationIO.ReadMigration(fm)
        read_back = {"llh": raw(rd.llh), "splitT": rd.splitT, "sampleDate": rd.sampleDate, "thrh": raw(rd.thrh), "jaf": [float(v) for v in rd.jaf],
                     "times": raw(rd.times), "lambda1": raw(rd.lambda1), "lambda2": raw(rd.lambda2), "lambdah1": raw(rd.lambdah1),
                     "lambdah2": raw(rd.lambdah2), "migStart": rd.migStart, "migEnd": rd.migEnd, "mi": rd.mi}
        writer.append({"name": name, "read_back": read_back,
                       "in": {"times": T, "lambdas": L, "sfs": S, "split": split, "mi": mi, "pu": pu, "params": params,
                                            "kw": kw, "thrh": [0.0521, 0.0104], "scaleTime": 1234.5, "scaleEPS": 0.75},
                       # attribute values as the reference holds them (Python ints stay ints: str(0) != str(0.0) in the text)
                       "model": {"times": raw(m.times), "splitT": m.splitT, "sampleDate": m.sampleDate, "thrh": raw(m.thrh),
                                 "JAFS": raw(m.JAFS), "dataJAFS": raw(m.dataJAFS), "lc": raw(m.lc), "lh": raw(m.lh),
                                 "mi": raw(m.mi), "Pr": raw(m.Pr), "llh": raw(m.llh)},
                       "text": text})
    # bootstrap resampling: the reference draws with the module-level `random` (seeded here; MiSTI.py seeds from the clock)
    import random as pyrandom
    rows = synth.chunk_rows([1000000, 52000, 31000, 47000, 28000, 9000, 14000, 8000], 20)
    boot = []
    for seed in (0, 1, 12345):
        for normalize in (False, True):
            pyrandom.seed(seed)
            draws = [migrationIO.BootstrapJAFS(types.SimpleNamespace(jafs=[list(r) for r in rows]), normalize) for _ in range(3)]
            boot.append({"seed": seed, "normalize": normalize, "draws": [[float(v) for v in d] for d in draws]})
    pyrandom.seed(5)
    table = [[sum(r[i] for r in rows) for i in range(8)]] + [migrationIO.BootstrapJAFS(types.SimpleNamespace(jafs=[list(r) for r in rows]), False) for _ in range(4)]
    # MiSTI.py end to end, one OS process per run (its module state - Units, the JAFS reader's mutable default - is global)
    f1, f2, fj = (os.path.join(tmp, n) for n in ("g1.psmc", "g2.psmc", "data.sfs"))
    t1, t2 = synth.psmc_text(16, 1, synth.THETA_1), synth.psmc_text(17, 2, synth.THETA_2)
    open(f1, "w").write(t1)
    open(f2, "w").write(t2)
    d = migrationIO.ReadPSMC(f1, f2, 0)
    truth = case("tmp", d.times, d.lambdas, [1] * 8, 20, trueEPS=True, cpfit=True, unfolded=True)
    from misti_amd import io as mio
    jtext = mio.format_jsfs(synth.chunk_rows(synth.counts_from_spectrum(truth["out"]["JAFS"], 200000), 5))
    open(fj, "w").write(jtext)
    runs = []
    for args in (["20", "--cpfit", "-bs", "0", "-o", "res.mi"], ["20"], ["19.5", "--cpfit", "-uf", "-bs", "2"],
                 ["20", "-mi", "1", "2", "20", "0.1", "0", "--cpfit", "-bs", "0", "-o", "res.mi"]):
        argv = ["MiSTI.py", "g1.psmc", "g2.psmc", "data.sfs"] + args + ["-wd", tmp]
        code = ("import numpy, sys, runpy; numpy.mat = numpy.asmatrix; sys.argv = %r; sys.path.insert(0, '/root/reference'); "
                "runpy.run_path('/root/reference/MiSTI.py', run_name='__main__')" % (argv,))
        res = os.path.join(tmp, "res.mi")
        if os.path.exists(res):
            os.remove(res)
        r = subprocess.run([sys.executable, "-c", code], cwd="/root/reference", capture_output=True, text=True,
                           env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("bs_id =")]
        assert len(line) == 1, r.stdout[-2000:]
        runs.append({"args": args, "result_line": line[0], "stdout": r.stdout.replace(tmp, "<wd>"),
                     "out_file": open(res).read() if os.path.exists(res) else None})
    return {"writer": writer, "bootstrap": {"rows": rows, "draws": boot, "table_seed5": [[float(v) for v in r] for r in table]},
            "cli": {"psmc1": t1, "psmc2": t2, "jsfs": jtext, "units": open("/root/reference/setunits.txt").read(), "runs": runs}}


def dedupe(cases):
    """Store each (times, lambdas) grid once; cases refer to it by key."""
    grids = {}
    for c in cases:
        key = None
        blob = json.dumps([c["in"]["times"], c["in"]["lambdas"]])
        for k, v in grids.items():
            if v["_blob"] == blob:
                key = k
        if key is None:
            key = "grid%d" % len(grids)
            grids[key] = {"_blob": blob, "times": c["in"]["times"], "lambdas": c["in"]["lambdas"]}
        del c["in"]["times"], c["in"]["lambdas"]
        c["in"]["grid"] = key
    for v in grids.values():
        del v["_blob"]
    return grids


def main():
    t0 = time.time()
    if "--host-only" in sys.argv:
        with tempfile.TemporaryDirectory() as tmp:
            host = host_fixtures(tmp)
        json.dump({"generator": "tests/golden/make_golden.py", **host}, open(os.path.join(HERE, "golden_host.json"), "w"))
        return
    if "--campaign-only" in sys.argv:
        import gzip
        have, have_traces = None, {}
        if "--keep" in sys.argv:                     # cases (and traces) already in the file stay as they are; only new picks are run
            d = json.load(open(os.path.join(HERE, "golden_campaign.json")))
            have = {}
            for c in d["cases"]:
                if "grid" in c["in"]:
                    g = d["grids"][c["in"].pop("grid")]
                    c["in"]["times"], c["in"]["lambdas"] = g["times"], g["lambdas"]
                have[(c["campaign"]["seed"], c["campaign"]["model"], c["campaign"]["cand"])] = c
            have_traces = {t["name"]: t for t in json.load(gzip.open(os.path.join(HERE, "golden_campaign_traces.json.gz"), "rt"))["cases"]}
        cc = campaign_cases(have=have)
        cc_traces = [have_traces[c["name"]] if c["name"] in have_traces else traced(c) for c in cc if c["out"]["llh"] is not None]
        json.dump({"generator": "tests/golden/make_golden.py --campaign-only", "scipy": "1.15.3", "numpy": "2.2.6", "grids": dedupe(cc), "cases": cc},
                  open(os.path.join(HERE, "golden_campaign.json"), "w"))
        with gzip.open(os.path.join(HERE, "golden_campaign_traces.json.gz"), "wt") as f:
            json.dump({"generator": "tests/golden/make_golden.py --campaign-only", "scipy": "1.15.3", "cases": cc_traces}, f)
        print("wrote %d campaign cases in %.1f s" % (len(cc), time.time() - t0))
        return
    with tempfile.TemporaryDirectory() as tmp:
        readers = reader_dumps(tmp)
    json.dump({"generator": "tests/golden/make_golden.py", "cases": readers},
              open(os.path.join(HERE, "golden_readers.json"), "w"))
    a = anchors()
    traces = [traced(c) for c in a if wants_trace(c)]
    json.dump({"generator": "tests/golden/make_golden.py", "scipy": "1.15.3", "numpy": "2.2.6",
               "grids": dedupe(a), "cases": a},
              open(os.path.join(HERE, "golden_small.json"), "w"))
    s = synthetic_cases()
    traces += [traced(c) for c in s if wants_trace(c)]
    for c in s:                        # the pair-state trace is pinned by the small cases
        c["out"].pop("Pr", None)
    json.dump({"generator": "tests/golden/make_golden.py", "scipy": "1.15.3", "numpy": "2.2.6",
               "grids": dedupe(s), "cases": s},
              open(os.path.join(HERE, "golden_synthetic.json"), "w"))
    ms = ms_cases()
    json.dump({"generator": "tests/golden/make_golden.py", "scipy": "1.15.3", "numpy": "2.2.6", "cases": ms},
              open(os.path.join(HERE, "golden_ms.json"), "w"))
    sw = sweep_cases()
    for c in sw:
        c["out"].pop("Pr", None)
    sw_traces = [traced(c) for c in sw if c["name"] in ("sw_df_st20_mc9_r0", "sw_cp_st22.5_mc10_r1")]
    json.dump({"generator": "tests/golden/make_golden.py", "scipy": "1.15.3", "numpy": "2.2.6", "grids": dedupe(sw), "cases": sw},
              open(os.path.join(HERE, "golden_sweep.json"), "w"))
    with tempfile.TemporaryDirectory() as tmp:
        host = host_fixtures(tmp)
    json.dump({"generator": "tests/golden/make_golden.py", **host}, open(os.path.join(HERE, "golden_host.json"), "w"))
    import gzip
    with gzip.open(os.path.join(HERE, "golden_traces.json.gz"), "wt") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "scipy": "1.15.3", "cases": traces + sw_traces}, f)
    n_inf = sum(1 for x in a + s + sw if x["out"]["llh"] is None)
    print("wrote %d small + %d synthetic + %d sweep cases (%d -inf), %d solver traces in %.1f s"
          % (len(a), len(s), len(sw), n_inf, len(traces) + len(sw_traces), time.time() - t0))


if __name__ == "__main__":
    main()
