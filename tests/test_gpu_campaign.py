"""Randomised differential campaign as a test: 3 x 600 random models (numT 8-40, all flag combinations, bands in both
directions, pulses, ancient sample, fractional splits), each evaluated as ONE batch of 6-28 candidates through the C
ABI - so chains are shared and the trunk paths run - against the oracle's value of every candidate
(tests/golden/campaign_seed{1,2,3}.json.gz, written by `tools/random_campaign.py --make-ref`; 27 minutes of oracle time
each, plus tools/self_perturbation.py's studies of the oracle's own spread)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,n_expected", [(1, 7648), (2, 7694), (3, 7434)])
def test_random_batches_against_the_oracle(seed, n_expected):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import random_campaign as rc
    rng = np.random.default_rng(seed)
    cases = [rc.random_batch(rng) for _ in range(600)]
    n_jobs = sum(len(c["split"]) for c in cases)
    ref = rc.load_ref(os.path.join(GOLDEN, "campaign_seed%d.json.gz" % seed), 600, seed, n_jobs)
    rep = rc.compare(cases, ref)
    s = rep["stats"]
    assert s["candidates"] == n_jobs == n_expected
    # failure against value only where the reference itself flips under a 2^-48 perturbation
    assert s["status_mismatch"] <= 3, rep["bad"][:5]
    comparable = s["tight"] + s["self_bound"] + s["internal_bound"] + s["outside"]
    assert comparable >= n_expected - 700                  # the rest fails on both sides (negative rates, failed corrections)
    # the contract per candidate (tests/parity.py): 1e-9 (+ rounding floor), or 10 x the reference's own spread under 4-64
    # perturbations of 2^-48 for THAT candidate, or (the 162 candidates beyond twice that spread were studied) under one
    # ulp in its own pair-chain expm (tests/golden/campaign_seed1.json.gz, tools/self_perturbation.py)
    assert s["tight"] >= 0.70 * comparable
    # Discrete stop/continue flips of SciPy's tests are rare events of the reference too (a few per thousand
    # ill-conditioned solves): a candidate whose flip the 4-64 perturbed reference runs did not happen to sample lands
    # outside.  The fixture was studied against one build; another rounding realisation moves which candidates those are,
    # so a small number is tolerated here - all of them in the noise-driven class - and profiles/ reports the exact count.
    assert s["outside"] <= comparable // 150, rep["outside"][:10]
    for rel, spread, idx, ci, k, split, run, cpfit, kinds, internal in rep["outside"]:
        c = cases[ci]
        default_mig = (not c["flags"]["cpfit"]) and (len(c["bands"]) > 0 or len(c["pulses"]) > 0)
        assert run >= 5.0 or default_mig or rel <= 1e-8, (idx, rel, spread, run)
        assert rel <= 5e-2, (idx, rel)
