"""The compiled CPU baseline (oracle/cpu/misti_cpu.cpp: C++17 + OpenMP restatement of the reference's algorithm, a
reported baseline and second checker - never part of the product) against the reference-generated golden vectors,
under the same per-case contract as the HIP path (tests/parity.py)."""
import math

import numpy as np
import pytest

from conftest import load_golden
from parity import SELF_FACTOR, determined, internal_of, llk_bound, spread_of, wide_of

BASELINE_FACTOR = 10.0

CASES = [c for f in ("golden_small", "golden_synthetic", "golden_sweep", "golden_campaign") for c in load_golden(f)]


def to_abi(case):
    """Golden-case input -> the band / pulse descriptors of the C interfaces (params in option order)."""
    i = case["in"]
    split = i["split"]
    bands, pulses, k = [], [], 0
    for pop, start, end, val, opt in i["mi"]:
        bands.append((int(pop) - 1, int(start), int(end), float(val), k if int(opt) == 1 else -1))
        k += int(opt) == 1
    for pop, t, val, opt in i["pu"]:
        pulses.append((int(pop) - 1, int(t), float(val), k if int(opt) == 1 else -1))
        k += int(opt) == 1
    kw = i["kw"]
    flags = dict(cpfit=bool(kw.get("cpfit")), true_eps=bool(kw.get("trueEPS")), smooth=bool(kw.get("smooth")), unfolded=bool(kw.get("unfolded")))
    return bands, pulses, k, flags, int(kw.get("sampleDate", 0)), float(kw.get("mixtureTH", 0.0)), split


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_compiled_baseline_against_the_reference(case):
    from oracle.cpu_baseline import cpu_eval
    i, o = case["in"], case["out"]
    bands, pulses, P, flags, sd, mth, split = to_abi(case)
    llk, jafs, status, run, _ = cpu_eval(i["times"], i["lambdas"], bands, pulses, flags, sd, [split], [i["params"]] if P else None, [i["sfs"]], P,
                                         mixture_th=mth, threads=1)
    if o["llh"] is None:
        want = {"Hit negative value of migration rate": 1, "Lambda correction failed": 2}[o["stdout"][0]]
        assert status[0] == want or o.get("pert_finite", 0) > 0, (status[0], o["stdout"])
        return
    if status[0] != 0:
        assert o.get("pert_fail", 0) > 0 or o.get("internal_fail", 0) > 0, (status[0], o["llh"])
        return
    bound, clause = llk_bound(o["llh"], i["sfs"], o["JAFS"], flags["unfolded"], spread_of(o), internal_of(o), wide_of(o))
    if clause != "1e-9":
        # The baseline is the first-pass CHECKER (its own Pade-13 expm and LU, its own rounding), not the product: it stays pinned at round 4's
        # factor 10 - at the product's factor 3 it misses one golden itself (camp_s4_m581_c14, two-way --cpfit runaway: 7.5 x the reference's
        # spread, where the HIP path's closed-form exponential is inside it).  Outliers of a first pass are judged by the reference, never by it.
        bound *= BASELINE_FACTOR / SELF_FACTOR
    assert abs(llk[0, 0] - o["llh"]) <= bound, (llk[0, 0], o["llh"], bound, clause)
    if determined(o):
        np.testing.assert_allclose(jafs[0], o["JAFS"], rtol=1e-8)


def test_batch_and_threads_change_nothing():
    """Candidates are independent tasks: a batch on several threads gives what one call per candidate gives."""
    from oracle.cpu_baseline import cpu_eval
    c = next(c for c in CASES if c["name"] == "c1_n32_cpfit")
    i = c["in"]
    bands, pulses, P, flags, sd, mth, _ = to_abi(c)
    splits = [18.0, 19.5, 20.0, 22.25, 25.0]
    rows = [i["sfs"], [v * 0.5 for v in i["sfs"]]]
    a = cpu_eval(i["times"], i["lambdas"], bands, pulses, flags, sd, splits, None, rows, P, threads=4)
    for k, s in enumerate(splits):
        b = cpu_eval(i["times"], i["lambdas"], bands, pulses, flags, sd, [s], None, rows, P, threads=1)
        assert np.array_equal(a[0][k], b[0][0]) and a[2][k] == b[2][0]
    assert (a[2] == 0).all() and np.isfinite(a[0]).all()
