"""The mode-aware reading of clause 2 (tests/parity.py: modes_of / branch_of / minority_tail), on the CPU: the classification itself and
the reference's own branch structure on the fixture that holds EVERY chain of a held-out grid (golden_config2b_allchains.json)."""
import numpy as np

from conftest import load_golden
from parity import MODE_RTOL, branch_of, chain_key, minority_tail, modes_of, reference_runs


def test_flips_are_modes_and_conditioning_is_not():
    a, b = -171413.834289, -171414.708228
    runs = [a * (1 + 1e-11 * k) for k in range(61)] + [b * (1 + 1e-11 * k) for k in range(3)]
    m = modes_of(runs)
    assert [len(c) for c in m] == [61, 3]
    spread = [-1000.0 * (1 + 1e-7 * k / 32.0) for k in range(33)]            # a llh with a condition number of 1e8: every run a little different
    assert len(modes_of(spread)) == 1
    assert len(modes_of([-5.0, -5.0 * (1 + MODE_RTOL / 2)])) == 1
    out = dict(llh=a, pert_llh=runs)
    on_major, on_minor, nowhere = branch_of(out, a), branch_of(out, b), branch_of(out, 0.5 * (a + b))
    assert on_major["mode"] == 0 and abs(on_major["share"] - 62 / 65) < 1e-12
    assert on_minor["mode"] == 1 and abs(on_minor["share"] - 3 / 65) < 1e-12
    assert nowhere["mode"] is None and nowhere["share"] == 0.0 and nowhere["n_modes"] == 2
    assert branch_of(dict(llh=a), a) is None                                    # no run lists: nothing to classify


def test_poisson_binomial_tail():
    assert minority_tail([], 0) == 1.0 and minority_tail([0.5], 1) == 0.5
    assert abs(minority_tail([0.05] * 20, 1) - (1 - 0.95 ** 20)) < 1e-12
    p = [0.1, 0.2, 0.3]
    brute = sum(np.prod([q if (k >> i) & 1 else 1 - q for i, q in enumerate(p)]) for k in range(8) if bin(k).count("1") >= 2)
    assert abs(minority_tail(p, 2) - brute) < 1e-12


def test_every_chain_of_the_held_out_grid():
    """64 chains fixed in advance, base + 64 + 16 reference runs each: about half are bimodal, and the reference's own base run is off its
    majority branch on a few of them - the rate an independent implementation is allowed."""
    cases = load_golden("golden_config2b_allchains")
    assert len(cases) == 64 and len({chain_key(c) for c in cases}) == 64
    bimodal, expected, base_off = 0, 0.0, 0
    for c in cases:
        o = c["out"]
        assert o["llh"] is not None and len(reference_runs(o)) >= 60           # base + 64 input perturbations + the one-ulp-in-expm runs that found a value
        b = branch_of(o, o["llh"])
        if b["n_modes"] >= 2:
            bimodal += 1
            expected += 1.0 - b["majority_share"]
            base_off += b["mode"] != 0
    assert 25 <= bimodal <= 40 and 4.0 <= expected <= 7.0 and 1 <= base_off <= 6, (bimodal, expected, base_off)
