"""Oracle-free invariants on random models (tools/random_campaign.py's generator, another seed): how a candidate is
batched and which of the sharing shortcuts run must never change a bit of its result.

  * trunk on / off (MISTI_NO_TRUNK), the trunk following its chain or placed after it (MISTI_NO_FOLLOW);
  * a candidate alone in a batch vs inside the batch it came with (different launch shapes, chain sharing, dispatch order);
  * the same batch twice on one context (launch shape taken from the first batch's chain count) and candidates permuted.
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def same(a, b, rows_a=None, rows_b=None):
    ia = slice(None) if rows_a is None else rows_a
    ib = slice(None) if rows_b is None else rows_b
    return (np.array_equal(a.status[ia], b.status[ib]) and np.array_equal(a.llk[ia], b.llk[ib], equal_nan=True)
            and np.array_equal(a.jafs[ia], b.jafs[ib], equal_nan=True) and np.array_equal(a.lc[ia], b.lc[ib], equal_nan=True))


def test_batching_and_sharing_never_change_a_result():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import random_campaign as rc
    from misti_amd.engine import Engine
    rng = np.random.default_rng(987654321)
    n_models = n_cand = n_ok = 0
    for _ in range(160):
        c = rc.random_batch(rng)
        n = len(c["split"])
        make = lambda: Engine(c["times"], c["lh"], c["bands"], c["pulses"], n_param=c["P"], sample_date=c["sd"], **c["flags"])
        ev = lambda e, idx=None: e.evaluate(c["split"] if idx is None else c["split"][idx],
                                            (None if c["params"] is None else (c["params"] if idx is None else c["params"][idx])),
                                            [c["sfs"]], want_lc=True)
        with make() as e:
            base = ev(e)
            again = ev(e)
            perm = rng.permutation(n)
            shuffled = ev(e, perm)
            pick = rng.choice(n, size=min(3, n), replace=False)
            singles = [ev(e, np.array([k])) for k in pick]
        assert same(base, again), c
        assert same(base, shuffled, perm, None), c
        for k, s in zip(pick, singles):
            assert same(base, s, np.array([k]), None), (c, k)
        for env in ({"MISTI_NO_TRUNK": "1"}, {"MISTI_NO_FOLLOW": "1"}):
            os.environ.update(env)
            try:
                with make() as e:
                    other = ev(e)
            finally:
                for k in env:
                    os.environ.pop(k, None)
            assert same(base, other), (c, env)
        n_models += 1
        n_cand += n
        n_ok += int((base.status == 0).sum())
    assert n_models == 160 and n_cand > 1500 and n_ok > n_cand // 2


def test_time_rescaling_and_population_swap_on_random_models():
    """Two symmetries of the model that no implementation detail shares between the two evaluations:
    rescaling time by a power of two (interval lengths x a, every rate / a) leaves the normalised spectrum unchanged;
    swapping the populations permutes its classes (0100<->0001, 1100<->0011, 1101<->0111).  Random models with all
    flag combinations, bands, pulses, fractional splits (sample date 0: an ancient second genome breaks the swap)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import random_campaign as rc
    from misti_amd.engine import Engine
    rng = np.random.default_rng(24680)
    perm = [2, 5, 0, 3, 6, 1, 4]
    n_scale = n_swap = n_models = 0
    worst_scale = worst_swap = 0.0
    while n_models < 120:
        c = rc.random_batch(rng)
        if c["sd"] != 0:
            continue
        n_models += 1
        noisy_fit = (not c["flags"]["cpfit"]) and (not c["flags"]["true_eps"]) and (len(c["bands"]) > 0 or len(c["pulses"]) > 0)
        kw = dict(n_param=c["P"], sample_date=0, **c["flags"])
        with Engine(c["times"], c["lh"], c["bands"], c["pulses"], **kw) as e:
            base = e.evaluate(c["split"], c["params"], [c["sfs"]])
        a = 4.0
        par_scaled = None
        if c["P"]:
            par_scaled = c["params"].copy()
            for (pop, s, en, v, par) in c["bands"]:
                if par >= 0:
                    par_scaled[:, par] /= a                      # rates scale, pulse fractions do not
        bands_scaled = [(pop, s, en, v / a, par) for (pop, s, en, v, par) in c["bands"]]
        with Engine(np.array(c["times"]) * a, np.array(c["lh"]) / a, bands_scaled, c["pulses"], **kw) as e:
            scaled = e.evaluate(c["split"], par_scaled, [c["sfs"]])
        bands_sw = [(1 - pop, s, en, v, par) for (pop, s, en, v, par) in c["bands"]]
        pulses_sw = [(1 - pop, t, v, par) for (pop, t, v, par) in c["pulses"]]
        with Engine(c["times"], np.array(c["lh"])[:, ::-1].copy(), bands_sw, pulses_sw, **kw) as e:
            swapped = e.evaluate(c["split"], c["params"], [c["sfs"]])
        for k in range(len(c["split"])):
            determined = (not noisy_fit) and base.status[k] == 0 and base.runaway[k] < 5.0
            if not determined:
                continue
            assert scaled.status[k] == 0 and swapped.status[k] == 0, (c, k)
            d2 = np.max(np.abs(swapped.jafs[k][perm] / base.jafs[k] - 1))
            worst_swap = max(worst_swap, d2)
            n_swap += 1
            assert d2 < 1e-9, (c, k, d2)
            if c["flags"]["cpfit"]:
                # only --cpfit is scale-free: the default fit stops SciPy's solver on an ABSOLUTE gradient tolerance of a
                # residual in time units (expected coalescence time), also in the post-split refit that runs with trueEPS
                d1 = np.max(np.abs(scaled.jafs[k] / base.jafs[k] - 1))
                worst_scale = max(worst_scale, d1)
                n_scale += 1
                assert d1 < 1e-9, (c, k, d1)
    print("time rescaling worst %.3g over %d candidates, population swap worst %.3g over %d" % (worst_scale, n_scale, worst_swap, n_swap))
    assert n_scale > 300 and n_swap > 400
