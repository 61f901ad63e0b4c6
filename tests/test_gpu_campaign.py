"""Randomised differential campaign as a test: 600 random models (numT 8-40, all flag combinations, bands in both
directions, pulses, ancient sample, fractional splits), each evaluated as ONE batch of 6-28 candidates through the C
ABI - so chains are shared and the trunk paths run - against the oracle's value of every candidate
(tests/golden/campaign_seed1.json.gz, written by `tools/random_campaign.py --make-ref`; 27 minutes of oracle time)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def test_random_batches_against_the_oracle():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import random_campaign as rc
    rng = np.random.default_rng(1)
    cases = [rc.random_batch(rng) for _ in range(600)]
    n_jobs = sum(len(c["split"]) for c in cases)
    ref = rc.load_ref(os.path.join(GOLDEN, "campaign_seed1.json.gz"), 600, 1, n_jobs)
    rep = rc.compare(cases, ref)
    s = rep["stats"]
    assert s["candidates"] == n_jobs == 7648
    assert s["status_mismatch"] == 0 and not rep["bad"], rep["bad"][:5]
    assert s["regular"] >= 4000
    # the 1e-9 contract (+ rounding floor) on every determined candidate, up to stop/continue flips of SciPy's gtol test
    assert s["regular_within_tol"] >= s["regular"] - max(2, s["regular"] // 500)
    assert rep["worst_regular"] <= 1e-7
    # noise-driven class (runaway corrected rate; default fit with a band or a pulse): the reference itself moves by
    # 1e-8..1e-1 under a 2^-48 perturbation of its inputs there - only gross agreement is meaningful
    big = [b for b in rep["loose"] if b[0] > 1e-3]
    assert len(big) <= len(rep["loose"]) // 100
    assert all(b[4] >= 5.0 for b in big)            # and only where the corrected rate ran away
