"""Randomised differential campaign as a test: 6 x 600 random models (numT 8-40, all flag combinations, bands in both directions, pulses,
ancient sample, fractional splits), each evaluated as ONE batch of 6-28 candidates through the C ABI - so chains are shared and the trunk
paths run - against the oracle's value of every candidate (tests/golden/campaign_seed{1..5}.json.gz: tools/random_campaign.py --make-ref).

Round 4's protocol, fixed before the device was consulted (tools/uniform_spread.py): EVERY candidate of the noise class - corrected rate x
interval length >= 5, default fit with a band or pulse, or "correction failed" in the oracle - has exactly 16 runs on inputs perturbed by
2^-48 and 16 runs with one ulp of noise in the pair chain's expm (compiled baseline); `spread` = the largest relative change of the llh.
Nothing is deepened afterwards.  Seed 5 was generated in round 4 and is the held-out fixture.  FIRST-PASS result (profiles/
r04_random_campaign_seed*.txt): 20 of 35 850 comparable candidates outside (6 of them in seed 5), no status mismatch.  Each of the 20 was
then run through /root/reference ITSELF with 64 input perturbations, 16 one-ulp-in-expm and 16 one-ulp-in-residual runs
(tests/golden/golden_campaign.json, tests/test_gpu_golden.py::test_campaign_worst): all 20 lie within the reference's own spread (the
reference reaches the device's value in its perturbed runs) - 9 of them are default-fit candidates WITHOUT migration, 1.0e-9 ... 2.3e-9 off,
which the class definition above leaves out and the reference itself moves by 1.1e-9 ... 3.8e-9.  Seed 6 - generated after all that, with
every default-fit candidate in the class - has NO candidate outside in its first pass (6 805 comparable) and no status mismatch."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from parity import record

pytestmark = pytest.mark.gpu


# First pass, measured on MI355X with this round's build: candidates, comparable ones, within 1e-9, and the candidates OUTSIDE the
# contract under the fixture's uniform spreads (candidate index -> measured relative distance), each of which is a reference-run golden.
MEASURED = {1: dict(n=7648, comparable=7103, tight=5249, outside={5600: 1.75e-9, 5601: 1.88e-9, 5602: 2.04e-9, 5603: 2.30e-9, 5604: 1.23e-9}),
            2: dict(n=7694, comparable=7080, tight=5379, outside={464: 1.85e-8, 466: 1.94e-8, 472: 2.38e-8, 476: 2.70e-8, 478: 2.64e-8, 480: 2.56e-8}),
            3: dict(n=7434, comparable=6811, tight=5145, outside={3137: 1.62e-9}),
            4: dict(n=7579, comparable=6779, tight=5097, outside={7331: 9.38e-8, 3867: 1.03e-9}),
            5: dict(n=7561, comparable=6964, tight=5406, outside={3642: 1.87e-6, 559: 6.45e-8, 560: 6.54e-8, 561: 6.96e-8, 6935: 1.13e-9, 5877: 1.01e-9}),
            # second held-out fixture, generated after everything above: the noise class now includes EVERY default-fit candidate (class version 2:
            # the reference-run studies of seeds 1-5 showed the default fit without migration determined to 1e-9 ... 4e-9 only).  First pass: none outside.
            6: dict(n=7372, comparable=6805, tight=5112, outside={})}


def studied():
    """(seed, model, candidate) of every campaign candidate that /root/reference itself was run on."""
    d = json.load(open(os.path.join(GOLDEN, "golden_campaign.json")))
    return {(c["campaign"]["seed"], c["campaign"]["model"], c["campaign"]["cand"]) for c in d["cases"]}


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_batches_against_the_oracle(seed):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import random_campaign as rc
    want = MEASURED[seed]
    rng = np.random.default_rng(seed)
    cases = [rc.random_batch(rng) for _ in range(600)]
    n_jobs = sum(len(c["split"]) for c in cases)
    ref = rc.load_ref(os.path.join(GOLDEN, "campaign_seed%d.json.gz" % seed), 600, seed, n_jobs)
    rep = rc.compare(cases, ref)
    s = rep["stats"]
    record("campaign_seed%d" % seed, **{k: int(v) for k, v in s.items()}, outside_list=[(int(b[2]), float(b[0])) for b in rep["outside"]],
           outside_detail=[dict(idx=int(b[2]), model=int(b[3]), cand=int(b[4]), rel=float(b[0]), spread=float(b[1]), factor=float(b[0] / b[1]) if b[1] else None,
                                run=float(b[6]), cpfit=bool(b[7])) for b in rep["outside"]])
    assert s["candidates"] == n_jobs == want["n"]
    # a failure against a value only where the reference (its restatement) itself flips in its 32 runs: none measured, none allowed
    assert s["status_mismatch"] == 0, rep["bad"][:5]
    comparable = s["tight"] + s["self_bound"] + s["internal_bound"] + s["outside"]
    assert comparable >= want["comparable"] - 5           # the rest fails on both sides (negative rates, failed corrections) or is a reference flip
    assert s["tight"] >= want["tight"] - comparable // 100, (s["tight"], want["tight"])
    # OUTSIDE under the first-pass spreads: only candidates that have been run through the reference itself (the golden test holds each of
    # them to the reference's own value and spread), each no farther than measured
    ref_run = studied()
    for rel, spread, idx, ci, k, split, run, cpfit, kinds, internal in rep["outside"]:
        assert (seed, ci, k) in ref_run, "seed %d candidate %d (model %d cand %d): outside the contract (rel %.3g) and never run through the reference" % (seed, idx, ci, k, rel)
        assert rel <= 1.5 * want["outside"].get(idx, 1e-9), (idx, rel, spread, run)
