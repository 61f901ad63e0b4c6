"""The per-candidate contract of tests/parity.py at BASELINE's full sizes, for EVERY candidate: config 2 (the headline
grid, 4 096 candidates) under --cpfit AND under the reference's default fit, a 4 096-candidate sample of config 5 (65 536: ancient
sample, band x pulse) and of config 3 (16 384 random two-band parameter vectors).  The full grids of configs 3 and 5 (and config 3 under
the default fit) are checked the same way by tools/fullsize_report.py: profiles/r04_fullsize_contract.txt.

First pass - the checker is the compiled CPU baseline (oracle/cpu/misti_cpu.cpp - the reference's algorithm, dense expm + inverse and
SciPy's TRF restated, pinned on the reference's golden cases): fast enough (~100 candidates/s/core) to evaluate the whole grid AND, for
every candidate the HIP path does not match to 1e-9, that candidate's own spread under SIXTEEN 2^-48 perturbations of the inputs and
SIXTEEN runs with one ulp of noise in its pair-chain expm (the same depth for all of them, fixed in advance), at test time.
Second pass - every candidate the first pass leaves outside must be one that /root/reference ITSELF has been run on (64 input
perturbations + 16 one-ulp-in-expm + 16 one-ulp-in-residual runs: tests/golden/golden_fullsize.json / golden_default_fit.json,
tests/golden/make_fullsize.py) and must lie within the contract THERE (tests/test_gpu_golden.py checks each of them against the
reference's own value and spread).  An outlier without a reference-run study fails the test.

Round 4's finding behind the second pass: where device and baseline disagree beyond the baseline's spread, the reference sides with the
DEVICE about as often as with the baseline (config 3: all 12 studied candidates, e.g. start 6208 - device 4e-12 from the reference,
baseline 1.3e-9), and where it sides with the baseline its own perturbed runs reach the device's value (config 5: 31 of 31)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from parity import KNOWN_OUTSIDE, KNOWN_STATUS, status_flips_wide, baseline_contract, pinned, record

pytestmark = pytest.mark.gpu

# measured on MI355X with this round's build (profiles/r05_measured_guards.jsonl; round 4: profiles/r04_fullsize_contract.txt):
# candidates with a value on both sides, those within 1e-9, FIRST-PASS outside (against the compiled baseline, SELF_FACTOR = 3: round 4's
# factor 10 gave 0 / 10 / 0 / 1).  Every first-pass outlier carries a reference-run study and lies inside the contract against the REFERENCE.
MEASURED = {"config2": dict(both=4096, tight=3584, outside=0),
            "config2:default": dict(both=3264, tight=6, outside=19),       # before the stall rule (misti_kernels.hip: correct_body): tight 0, outside 23
            # the HELD-OUT instance of the headline grid (workloads.config2b: other PSMC curves, another true history; made at the end of round 5)
            "config2b": dict(both=4096, tight=3359, outside=15),           # one chain (rate index 35), rate x length 509 ... 672: reference-studied, golden_config2b.json
            # ... and the SECOND held-out instance, made after the stall rule (workloads.config2c): the whole grids by tools/fullsize_report.py - --cpfit 4 096 / 3 512 tight /
            # 20 flagged (all <= 0.63 x the reference's spread), default fit 3 254 / 3 tight / 0 outside / 4 status cases (reference flips)
            "config2c/4": dict(both=1024, tight=None, outside=21), "config2c:default/4": dict(both=813, tight=None, outside=0),
            # held-out instances of configs 3 and 5 (workloads.config3b / config5b): every fourth start / every 32nd candidate here; wider samples by tools/fullsize_report.py:
            # config3b 4 069 / 3 638 tight / 7 flagged (3 of them outside against the reference: KNOWN_OUTSIDE), config3b default 1 442 / 0 outside / 1 status case (a value in
            # the reference after all), config5b 7 968 / 5 271 tight / 0 outside, config5b default 3 520 / 0 outside
            "config3b/4": dict(both=4069, tight=3638, outside=7), "config5b/32": dict(both=None, tight=None, outside=0),
            "config2b:default/2": dict(both=1715, tight=None, outside=0),  # every second candidate; the whole grid (tools/fullsize_report.py): 3 429 / 5 tight / 0 outside - 270 outside before the stall rule
            "config5/16": dict(both=4080, tight=3168, outside=0),
            "config3/4": dict(both=4078, tight=3703, outside=4)}        # starts 1300, 8868, 9412, 13340: the device 4e-12 ... 2e-9 from the REFERENCE


def studied(workload):
    """Candidates of `workload` that /root/reference itself was run on (candidate -> case)."""
    from conftest import load_golden
    out = {}
    for f in ("golden_fullsize", "golden_default_fit", "golden_default_fit_256", "golden_fullsize_r05", "golden_config2b", "golden_config2c", "golden_config3b", "golden_config2n255", "golden_config2u", "golden_config2m", "golden_config2f"):
        if os.path.exists(os.path.join(GOLDEN, f + ".json")):
            for c in load_golden(f):
                if c["fullsize"]["workload"] == workload:
                    out[int(c["fullsize"]["cand"])] = c
    return out


def full_contract(w, idx, threads=16, kinds=16, internal=16):
    from misti_amd.engine import Engine
    split = w.split_time[idx]
    par = None if w.params is None else w.params[idx]
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        r = e.evaluate(split, par, w.jsfs[:1])
    sub = type(w)(w.name, w.times, w.lh, w.bands, w.pulses, w.n_param, w.flags, w.sample_date, split, par, w.truth, w.jsfs)
    rep = baseline_contract(sub, np.arange(len(idx)), r.llk, r.status, threads=threads, kinds=kinds, internal=internal)
    rep["hip_llk"], rep["hip_status"] = r.llk[:, 0], r.status
    return rep


def check(key, workload, idx, rep):
    """Guards = the measured counts (-1 % on the tight count); every first-pass outlier and every status mismatch must carry a
    reference-run study, under which the device's value (or status) is within the contract."""
    from parity import SELF_FACTOR, llk_tol
    want = MEASURED[key]
    record("fullsize_" + key, both=rep["both"], tight=rep["tight"], self_bound=rep["self_bound"],
           outside=[(int(idx[k]), float(rep["rel"][k])) for k in rep["outside"]], mismatch=[int(idx[k]) for k in rep["mismatch"]])
    if want["tight"] is not None:
        pinned(rep["tight"] >= want["tight"] - rep["both"] // 100, ("tight", key, rep["tight"], want))
    ref = studied(workload)
    for k in list(rep["outside"]) + list(rep["mismatch"]):
        cand = int(idx[k])
        assert cand in ref, "candidate %d of %s: outside the contract against the compiled baseline (rel %.3g) and never run through the reference" % (cand, workload, rep["rel"][k])
        o = ref[cand]["out"]
        h, hs = rep["hip_llk"][k], rep["hip_status"][k]
        if o["llh"] is None or hs != 0:
            # a failure against a value: only where the reference itself flips under its perturbed runs
            flips = (o.get("pert_finite", 0) > 0 or o.get("internal_finite", 0) > 0) if o["llh"] is None else (o.get("pert_fail", 0) > 0 or o.get("internal_fail", 0) > 0)
            flips = flips or status_flips_wide(ref[cand]["name"])       # the reference's own runs at 2^-40 ... 2^-32 (tests/parity.py)
            flips = flips or ref[cand]["name"] in KNOWN_STATUS           # the three listed expected failures (their own test: test_gpu_golden.py)
            assert (o["llh"] is None) == (hs != 0) or flips, (cand, o["llh"], hs)
            continue
        tol = max(1e-9 * abs(o["llh"]), SELF_FACTOR * max(o.get("spread") or 0.0, o.get("internal_spread") or 0.0, o.get("spread_wide") or 0.0) * abs(o["llh"]))
        if ref[cand]["name"] in KNOWN_OUTSIDE:                 # the round's four expected failures (tests/parity.py), should the first pass flag one: its pinned distance
            tol = max(tol, KNOWN_OUTSIDE[ref[cand]["name"]] * abs(o["llh"]))
        assert abs(h - o["llh"]) <= tol, (cand, h, o["llh"], abs(h - o["llh"]) / abs(o["llh"]), o.get("spread"), o.get("internal_spread"))


def test_headline_grid_every_candidate():
    """BASELINE config 2 at full size: all 4 096 candidates of the 64 x 64 split x rate grid.
    Measured: 3 584 within 1e-9 (87.5 %; worst 5.3e-10), 512 within 10 x their spread, NONE outside."""
    from misti_amd import workloads
    from misti_amd.engine import truth_spectrum
    w = workloads.config2(lambda *a: truth_spectrum(*a))
    idx = np.arange(w.n_cand)
    rep = full_contract(w, idx)
    check("config2", "config2", idx, rep)
    assert rep["both"] == 4096 and rep["worst_tight"] <= 2e-9 and len(rep["outside"]) == 0 and len(rep["mismatch"]) == 0


def test_headline_grid_default_fit_every_candidate():
    """The headline grid under the reference's DEFAULT fit (MiSTI.py:86,213: --cpfit is opt-in; LambdaSystem, CorrectLambda.py:94-110,303)
    - what `bench.py --fit default` times - every candidate.  The reference's own value is determined to 1e-6 ... 1e-3 only on this
    grid (the conditional expected coalescence time of a short interval barely depends on the rate: golden_default_fit.json, 16 + 16
    reference runs per candidate), so no candidate is within 1e-9 of the compiled baseline and all but a handful are within 10 x
    their spread; the first-pass outliers are reference-studied (the device is 5 - 8 x closer to the reference there than the baseline is)."""
    from misti_amd import workloads
    from misti_amd.engine import truth_spectrum
    w = workloads.config2(lambda *a: truth_spectrum(*a), cpfit=False)
    idx = np.arange(w.n_cand)
    rep = full_contract(w, idx)
    check("config2:default", "config2:default", idx, rep)
    pinned(rep["both"] >= MEASURED["config2:default"]["both"] - 40, ("both", "config2:default", rep["both"]))
    pinned(len(rep["outside"]) <= MEASURED["config2:default"]["outside"] + 2, ("first-pass outliers", "config2:default", len(rep["outside"])))


def test_held_out_grid_every_candidate():
    """`config2b`: the headline grid's shape on other data, made after everything was fixed (workloads.py) - every candidate, --cpfit."""
    from misti_amd import workloads
    from misti_amd.engine import truth_spectrum
    w = workloads.config2b(lambda *a: truth_spectrum(*a))
    idx = np.arange(w.n_cand)
    rep = full_contract(w, idx)
    check("config2b", "config2b", idx, rep)
    assert rep["both"] == 4096 and rep["worst_tight"] <= 2e-9 and len(rep["mismatch"]) == 0
    pinned(len(rep["outside"]) <= MEASURED["config2b"]["outside"] + 2, ("first-pass outliers", "config2b", len(rep["outside"])))       # each of them is held to its reference-run study by check()


def test_held_out_grid_default_fit():
    """... and under the reference's default fit, every second candidate: NONE outside at factor 3 against the compiled baseline.  (Without the stall rule: 270 of
    the grid's 3 430, and 3 of 8 sampled through the reference beyond 3 x its own spread - what this grid was made to find.)"""
    from misti_amd import workloads
    from misti_amd.engine import truth_spectrum
    w = workloads.config2b(lambda *a: truth_spectrum(*a), cpfit=False)
    idx = np.arange(0, w.n_cand, 2)
    rep = full_contract(w, idx)
    check("config2b:default/2", "config2b:default", idx, rep)
    assert len(rep["mismatch"]) == 0
    pinned(rep["both"] >= MEASURED["config2b:default/2"]["both"] - 20 and len(rep["outside"]) <= 2, ("both / first-pass outliers", "config2b:default/2", rep["both"], len(rep["outside"])))


@pytest.mark.parametrize("cpfit", [True, False])
def test_second_held_out_grid(cpfit):
    """`config2c`, every fourth candidate, both fits: every flagged candidate carries its reference-run study (golden_config2c.json) and lies inside the contract there."""
    from misti_amd import workloads
    from misti_amd.engine import truth_spectrum
    w = workloads.config2c(lambda *a: truth_spectrum(*a), cpfit=cpfit)
    idx = np.arange(3, w.n_cand, 4)              # offset 3: the flagged candidates of the whole grid sit on odd rate columns
    rep = full_contract(w, idx)
    key = "config2c/4" if cpfit else "config2c:default/4"
    check(key, "config2c" if cpfit else "config2c:default", idx, rep)
    pinned(len(rep["outside"]) <= MEASURED[key]["outside"] + 1, ("first-pass outliers", key, len(rep["outside"])))


def test_held_out_config3b_and_config5b():
    """Held-out instances of configs 3 and 5 (--cpfit): config3b every fourth start - the flagged ones are reference-studied (three of them outside there: their own expected
    failures in test_gpu_golden.py, accepted here at their pinned distance) -, config5b every 32nd candidate: none outside."""
    from misti_amd import workloads
    from misti_amd.engine import truth_spectrum
    w = workloads.config3b(lambda *a: truth_spectrum(*a))
    idx = np.arange(0, w.n_cand, 4)
    rep = full_contract(w, idx)
    check("config3b/4", "config3b", idx, rep)
    assert len(rep["mismatch"]) == 0
    pinned(len(rep["outside"]) <= MEASURED["config3b/4"]["outside"] + 2, ("first-pass outliers", "config3b/4", len(rep["outside"])))
    w = workloads.config5b(lambda *a: truth_spectrum(*a))
    idx = np.arange(0, w.n_cand, 32)
    rep = full_contract(w, idx)
    check("config5b/32", "config5b", idx, rep)
    assert len(rep["outside"]) == 0 and len(rep["mismatch"]) == 0


@pytest.mark.parametrize("name,stride", [("config2n64", 2), ("config2n255", 8)])
@pytest.mark.parametrize("cpfit", [True, False])
def test_held_out_other_grid_sizes(name, stride, cpfit):
    """Held-out grids at numT = 64 and numT = 255 (the C ABI's largest), both fits, strided samples (the whole grids: tools/fullsize_report.py, profiles/r05_fullsize_contract_first_pass.txt):
    nothing outside; status cases only where reference-studied (numT 255, default fit: one chain, the reference flips on all 30)."""
    from misti_amd import workloads
    from misti_amd.engine import truth_spectrum
    w = getattr(workloads, name)(lambda *a: truth_spectrum(*a), cpfit=cpfit)
    idx = np.arange(5, w.n_cand, stride)
    rep = full_contract(w, idx)
    key = "%s%s/%d" % (name, "" if cpfit else ":default", stride)
    MEASURED.setdefault(key, dict(both=None, tight=None, outside=0))
    check(key, name + ("" if cpfit else ":default"), idx, rep)
    assert len(rep["outside"]) == 0


@pytest.mark.parametrize("cpfit", [True, False])
def test_held_out_pulse_grid(cpfit):
    """Held-out grid of a pulse model (workloads.config2pu: split x pulse fraction, a fixed band), every eighth candidate: nothing outside, no status case (the whole grid: 4 096 / 4 050
    candidates with a value on both sides, 0 / 0 outside - profiles/r05_fullsize_contract_first_pass.txt)."""
    from misti_amd import workloads
    from misti_amd.engine import truth_spectrum
    w = workloads.config2pu(lambda *a: truth_spectrum(*a), cpfit=cpfit)
    rep = full_contract(w, np.arange(1, w.n_cand, 8))
    assert rep["both"] >= 500 and len(rep["outside"]) == 0 and len(rep["mismatch"]) == 0


def test_held_out_grid_true_eps():
    """The held-out grid under --trueEPS (MigrationInference.py:74: no lambda-correction, the spectrum path alone): all 4 096 candidates within 1e-9 of the compiled baseline."""
    from misti_amd import workloads
    from misti_amd.engine import truth_spectrum
    w = workloads.config2t(lambda *a: truth_spectrum(*a))
    rep = full_contract(w, np.arange(w.n_cand))
    assert rep["both"] == 4096 and rep["tight"] == 4096 and rep["worst_tight"] <= 1e-10 and len(rep["outside"]) == 0 and len(rep["mismatch"]) == 0


@pytest.mark.parametrize("cpfit", [False, True])
def test_held_out_config4b(cpfit):
    """Held-out instance of config 4 (no migration; 256 split values, a third of them fractional): every candidate within 1e-9 of the compiled baseline under both fits."""
    from misti_amd import workloads
    from misti_amd.engine import truth_spectrum
    w = workloads.config4b(lambda *a: truth_spectrum(*a), cpfit=cpfit)
    idx = np.arange(w.n_cand)
    rep = full_contract(w, idx)
    assert rep["both"] == 256 and rep["tight"] == 256 and rep["worst_tight"] <= 1e-9 and len(rep["outside"]) == 0 and len(rep["mismatch"]) == 0


def test_config5_sample_every_candidate():
    """BASELINE config 5 (ancient second genome, band x pulse x split): 4 096 of the 65 536 candidates, evenly spaced."""
    from misti_amd import workloads
    from misti_amd.engine import truth_spectrum
    w = workloads.config5(lambda *a: truth_spectrum(*a))
    idx = np.arange(0, w.n_cand, 16)
    check("config5/16", "config5", idx, full_contract(w, idx))


def test_config3_sample_every_candidate():
    """BASELINE config 3 (two optimised bands, random parameter vectors): 4 096 of the 16 384 starts."""
    from misti_amd import workloads
    from misti_amd.engine import truth_spectrum
    w = workloads.config3(lambda *a: truth_spectrum(*a))
    idx = np.arange(0, w.n_cand, 4)
    check("config3/4", "config3", idx, full_contract(w, idx))


def test_config3_default_fit_sample_every_candidate():
    """BASELINE config 3 under the reference's default fit: 1 024 of the 16 384 starts (the whole grid: tools/fullsize_report.py config3:default,
    profiles/r04_fullsize_contract.txt - 10 912 with a value on both sides, 10 911 within 10 x their spread, 1 outside, 46 failure-against-value;
    all 47 reference-studied: golden_default_fit.json).  Every flagged candidate of the sample must carry its reference-run study."""
    from misti_amd import workloads
    from misti_amd.engine import truth_spectrum
    w = workloads.config3(lambda *a: truth_spectrum(*a), cpfit=False)
    idx = np.arange(0, w.n_cand, 16)
    MEASURED.setdefault("config3:default/16", dict(both=None, tight=None, outside=None))
    rep = full_contract(w, idx)
    check("config3:default/16", "config3:default", idx, rep)
    assert rep["both"] >= 600
