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
