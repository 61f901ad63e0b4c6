"""Randomised differential campaign as a test: 8 x 600 random models (numT 8-40, all flag combinations, bands in both directions, pulses,
ancient sample, fractional splits), each evaluated as ONE batch of 6-28 candidates through the C ABI - so chains are shared and the trunk
paths run - against the oracle's value of every candidate (tests/golden/campaign_seed{1..8}.json.gz: tools/random_campaign.py --make-ref).

The protocol, fixed before the device is consulted (tools/uniform_spread.py): EVERY candidate of the noise class - corrected rate x
interval length >= 5, default fit (seeds 1-5: with a band or pulse; from seed 6 on: every default-fit candidate), or "correction failed" in the
oracle - has exactly 16 runs on inputs perturbed by 2^-48 and 16 runs with one ulp of noise in the pair chain's expm (compiled baseline);
`spread` = the largest relative change of the llh.  Nothing is deepened afterwards.  Seeds 5, 6, 7 and 8 were each generated AFTER the studies of
the seeds before them (held out); seed 7 after round 5 set the contract's factor to 3, seed 8 with the round's last build in place.
FIRST PASS at factor 3 (profiles/r05_random_campaign_seed*.txt): 12 / 11 / 10 / 7 / 7 / 1 / 9 / 2 of ~6 900 comparable candidates per seed outside (whole
chains fall out together), no status mismatch.  SECOND PASS: every one of them is run through /root/reference ITSELF with 64 input perturbations,
16 one-ulp-in-expm and 16 one-ulp-in-residual runs (tests/golden/golden_campaign.json, tests/test_gpu_golden.py::test_campaign_worst holds each
to the reference's own value and spreads)."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from parity import pinned, record

pytestmark = pytest.mark.gpu


# First pass, measured on MI355X with this round's build: candidates, comparable ones, within 1e-9, and the candidates OUTSIDE the
# contract under the fixture's uniform spreads (candidate index -> measured relative distance), each of which is a reference-run golden.
MEASURED = {
            1: dict(n=7648, comparable=7103, tight=5248, outside={2697: 7.41e-08, 2967: 5.73e-06, 2969: 5.17e-06, 2971: 4.23e-06, 2973: 3.84e-06, 2975: 3.72e-06, 4522: 8.85e-06, 5600: 1.75e-09, 5601: 1.88e-09, 5602: 2.04e-09, 5603: 2.3e-09, 5604: 1.23e-09}),
            2: dict(n=7694, comparable=7080, tight=5375, outside={222: 6.35e-07, 464: 1.85e-08, 466: 1.94e-08, 472: 2.38e-08, 476: 2.7e-08, 478: 2.64e-08, 480: 2.56e-08, 484: 2.37e-08, 540: 6.97e-09, 1182: 5.85e-08, 2697: 3.48e-09}),
            3: dict(n=7434, comparable=6811, tight=5147, outside={1876: 1.57e-08, 1880: 1.43e-08, 1882: 1.3e-08, 1886: 8.02e-09, 1888: 6.22e-09, 1890: 3.83e-09, 1892: 3.17e-09, 3137: 1.62e-09, 5270: 4.39e-08, 5271: 3.56e-08}),
            4: dict(n=7579, comparable=6779, tight=5099, outside={3812: 2.32e-06, 3813: 1.44e-06, 3814: 8.22e-07, 3816: 6.06e-07, 3817: 5.99e-07, 3867: 1.03e-09, 7331: 3.54e-07}),
            5: dict(n=7561, comparable=6964, tight=5405, outside={559: 6.45e-08, 560: 6.54e-08, 561: 6.96e-08, 3642: 1.85e-06, 5114: 4.63e-09, 5877: 1.01e-09, 6935: 1.13e-09}),
            6: dict(n=7372, comparable=6805, tight=5111, outside={3735: 1.95e-05}),
            # third held-out fixture: generated in round 5 AFTER the factor went to 3, the closed-form exponential and every study above (class version 2)
            7: dict(n=7409, comparable=6845, tight=5301, outside={1179: 1.1e-08, 1181: 1.14e-08, 1183: 1.31e-08, 1185: 1.31e-08, 1187: 1.28e-08, 1189: 1.29e-08,
                                                                  1191: 1.15e-08, 3886: 5.26e-09, 5206: 8.92e-07}),
            # fourth held-out fixture: generated with the round's LAST build in place, nothing looked at before the first pass
            8: dict(n=7452, comparable=6921, tight=5227, outside={5711: 1.11e-08, 7250: 1.32e-09}),
}
# (round 4, factor 10: 5 / 6 / 1 / 2 / 6 / 0 outside.  Round 5, factor 3: 12 / 11 / 10 / 7 / 7 / 1 - whole chains fall out together: the members of
# seed 1 model 229, seed 3 model 140, seed 4 model 303 share one chain each.  Seed 6 was generated after round 4's studies, seed 5 before them.)


def studied():
    """(seed, model, candidate) of every campaign candidate that /root/reference itself was run on."""
    d = json.load(open(os.path.join(GOLDEN, "golden_campaign.json")))
    return {(c["campaign"]["seed"], c["campaign"]["model"], c["campaign"]["cand"]) for c in d["cases"]}


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
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
           outside_detail=[dict(idx=int(b[2]), model=int(b[3]), cand=int(b[4]), rel=float(b[0]), spread=None if b[1] is None else float(b[1]),
                                factor=float(b[0] / b[1]) if b[1] else None, run=None if b[6] is None else float(b[6]), cpfit=bool(b[7])) for b in rep["outside"]])
    assert s["candidates"] == n_jobs == want["n"]
    # a failure against a value only where the reference (its restatement) itself flips in its 32 runs: none measured, none allowed
    assert s["status_mismatch"] == 0, rep["bad"][:5]
    comparable = s["tight"] + s["self_bound"] + s["internal_bound"] + s["outside"]
    pinned(comparable >= want["comparable"] - 5, ("comparable", seed, comparable, want["comparable"]))      # the rest fails on both sides (negative rates, failed corrections) or is a reference flip
    pinned(s["tight"] >= want["tight"] - comparable // 100, ("tight", seed, s["tight"], want["tight"]))
    assert s["tight"] >= 0.6 * comparable, (s["tight"], comparable)          # the contract's own floor: most comparable candidates are within 1e-9
    # OUTSIDE under the first-pass spreads: only candidates that have been run through the reference itself (the golden test holds each of
    # them to the reference's own value and spread), each no farther than measured
    ref_run = studied()
    for rel, spread, idx, ci, k, split, run, cpfit, kinds, internal in rep["outside"]:
        assert (seed, ci, k) in ref_run, "seed %d candidate %d (model %d cand %d): outside the contract (rel %.3g) and never run through the reference" % (seed, idx, ci, k, rel)
        pinned(rel <= 1.5 * want["outside"].get(idx, 1e-9), ("outlier distance", seed, idx, rel, spread, run))      # the contract for it: tests/test_gpu_golden.py::test_campaign_worst
