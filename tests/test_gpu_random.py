"""Randomised differential test: small random models (all flags, bands in both directions,
pulses, ancient second genome, fractional splits) through the HIP path and the oracle."""
import warnings

import numpy as np
import pytest

from parity import llk_tol

pytestmark = pytest.mark.gpu


def random_case(rng):
    numT = int(rng.integers(5, 14))
    times = list(np.round(10 ** rng.uniform(-2.3, -0.5, numT - 1), 6))
    lh = [[float(np.round(10 ** rng.uniform(-0.3, 0.4), 4)), float(np.round(10 ** rng.uniform(-0.3, 0.4), 4))] for _ in range(numT)]
    sd = int(rng.integers(0, 3)) if rng.random() < 0.3 else 0
    s_int = int(rng.integers(max(1, sd), numT - 1))
    split = s_int + (float(np.round(rng.uniform(0.1, 0.9), 3)) if rng.random() < 0.3 and s_int <= numT - 3 else 0.0)
    n_two = s_int + (1 if split != s_int else 0)
    bands, pulses, P = [], [], 0
    for pop in (0, 1):
        if rng.random() < 0.6 and n_two - sd >= 1:
            start = int(rng.integers(sd, n_two))
            end = -1 if rng.random() < 0.5 else int(rng.integers(start + 1, n_two + 1))
            opt = rng.random() < 0.5
            bands.append((pop, start, end, float(np.round(10 ** rng.uniform(-2, 0.3), 4)), P if opt else -1))
            P += int(opt)
    if rng.random() < 0.4 and n_two - sd >= 1:
        opt = rng.random() < 0.5
        pulses.append((int(rng.integers(0, 2)), int(rng.integers(sd, n_two)), float(np.round(rng.uniform(0.02, 0.6), 3)), P if opt else -1))
        P += int(opt)
    flags = dict(cpfit=bool(rng.random() < 0.6), true_eps=bool(rng.random() < 0.15), smooth=bool(rng.random() < 0.7),
                 unfolded=bool(rng.random() < 0.5))
    params = [float(np.round(10 ** rng.uniform(-2, 0.2), 4)) for _ in range(P)]
    for (pop, t, v, par) in pulses:
        if par >= 0:
            params[par] = float(np.round(rng.uniform(0.02, 0.6), 3))
    sfs = [1e5] + [float(v) for v in rng.integers(50, 3000, 7)]
    return dict(times=times, lh=lh, sd=sd, split=float(split), bands=bands, pulses=pulses, P=P, flags=flags, params=params, sfs=sfs)


def test_random_models_against_oracle():
    """The contract of tests/parity.py per model: 1e-9 (+ rounding floor), else within 10 x the oracle's own spread under
    8 perturbations of 2^-48 of ITS inputs, computed here for exactly the models that need it."""
    from parity import SELF_FACTOR, perturbed
    from misti_amd.engine import Engine
    from oracle.batch import oracle_eval
    rng = np.random.default_rng(20240607)
    n_checked = n_tight = n_self = 0
    outside = []
    for _ in range(120):
        c = random_case(rng)
        with Engine(c["times"], c["lh"], c["bands"], c["pulses"], n_param=c["P"], sample_date=c["sd"], **c["flags"]) as e:
            r = e.evaluate([c["split"]], [c["params"]] if c["P"] else None, [c["sfs"]])

        def oracle(times, lh):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                return oracle_eval(times, lh, c["bands"], c["pulses"], c["flags"], c["sd"], c["split"], c["params"], [c["sfs"]])
        o_llk, o_jafs, o_st, run = oracle(c["times"], c["lh"])
        n_checked += 1
        if o_st != 0 or r.status[0] != 0:
            if (o_st != 0) != (r.status[0] != 0):
                # a failure against a value: only where the oracle itself flips under perturbation
                fails = [oracle(*perturbed(c["times"], c["lh"], k))[2] != 0 for k in range(8)]
                assert (o_st != 0 and not all(fails)) or (o_st == 0 and any(fails)), (c, o_st, r.status[0])
            continue
        err = abs(r.llk[0, 0] - o_llk[0])
        if err <= llk_tol(o_llk[0], c["sfs"], o_jafs, c["flags"]["unfolded"]):
            n_tight += 1
            np.testing.assert_allclose(r.jafs[0], o_jafs, rtol=1e-7)
            continue
        vals = [oracle(*perturbed(c["times"], c["lh"], k)) for k in range(8)]
        fin = [v[0][0] for v in vals if v[2] == 0]
        spread = max(abs(v - o_llk[0]) for v in fin) if fin else 0.0
        if err <= SELF_FACTOR * spread:
            n_self += 1
        else:
            outside.append((err / abs(o_llk[0]), spread / abs(o_llk[0]), run, c["flags"]))
    assert n_tight >= 60, (n_checked, n_tight, n_self)
    # a stop/continue flip the eight perturbed runs did not sample: rare, and only in the noise-driven regimes
    assert len(outside) <= 2, outside
    for rel, spread, run, flags in outside:
        assert rel <= 2e-2 and (run >= 5.0 or not flags["cpfit"]), outside
