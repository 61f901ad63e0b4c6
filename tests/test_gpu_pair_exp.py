"""The pair chain's exponential with migration in BOTH directions (pair_eigen, misti_kernels.hip: the closed form over the three
eigenvalues of the symmetrisable generator, CorrectLambda.py:55-62) against 50-digit arithmetic, through the forward map of the C ABI
(misti_forward_rates: one exp(M T) v per genome and interval, no solver): stiff generators - rate x length up to 3e5 - strong, weak and
lopsided migration, coinciding poles (d0 == d1), and the non-stiff neighbours that take the series."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def exact_states(times, lh, mu0, mu1, n_int):
    import mpmath as mp
    mp.mp.dps = 50
    v = [mp.matrix([1, 0, 0]), mp.matrix([0, 1, 0])]
    rows = []
    for t in range(n_int):
        T = mp.mpf(float(times[t]))
        l0, l1 = mp.mpf(float(lh[t][0])), mp.mpf(float(lh[t][1]))
        a, b = mp.mpf(float(mu0)), mp.mpf(float(mu1))
        M = mp.matrix([[-2 * a - l0, 0, b], [0, -2 * b - l1, a], [2 * a, 2 * b, -a - b]]) * T
        E = mp.expm(M)
        v = [E * v[0], E * v[1]]
        rows.append([v[0][0], v[1][0], v[0][1], v[1][1], v[0][2], v[1][2]])
    return rows


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_two_way_pair_exponential_against_50_digits(seed):
    from misti_amd.engine import Engine
    rng = np.random.default_rng(seed)
    worst = 0.0
    n_stiff = 0
    for trial in range(24):
        numT = 6
        times = list(10 ** rng.uniform(-1.5, 0.3, numT - 1))
        kind = trial % 6
        lh = [[float(10 ** rng.uniform(-1, 5.3)), float(10 ** rng.uniform(-1, 5.3))] for _ in range(numT)]
        mu0, mu1 = float(10 ** rng.uniform(-4, 0.6)), float(10 ** rng.uniform(-4, 0.6))
        if kind == 1:                                   # one rate ordinary, the other run away
            for r in lh:
                r[0] = float(rng.uniform(0.3, 3))
        if kind == 2:                                   # coinciding poles: 2 mu0 + l0 == 2 mu1 + l1 exactly
            mu1 = mu0
            for r in lh:
                r[1] = r[0]
        if kind == 3:                                   # lopsided migration
            mu1 = mu0 * 1e-4
        if kind == 4:                                   # strong migration, small rates: stiff through mu alone
            mu0, mu1 = float(rng.uniform(4, 40)), float(rng.uniform(4, 40))
            lh = [[float(rng.uniform(0.01, 2)), float(rng.uniform(0.01, 2))] for _ in range(numT)]
        bands = [(0, 0, numT - 1, mu0, -1), (1, 0, numT - 1, mu1, -1)]
        with Engine(times, lh, bands, [], n_param=0, cpfit=True) as e:
            r = e.forward_rates([float(numT - 1)], None, want_pr=True, hold_mu=False)
        want = exact_states(times, lh, mu0, mu1, numT - 1)
        got = r["pr"][0] if isinstance(r, dict) else r[1][0]
        for t in range(numT - 1):
            T = times[t]
            nb = max(2 * mu0 + lh[t][0], 2 * mu1 + lh[t][1], mu0 + mu1) * T
            n_stiff += nb > 6.0
            w = [float(x) for x in want[t]]
            scale = max(abs(x) for x in w)
            if scale < 1e-280:
                break
            err = max(abs(got[t + 1][j] - w[j]) for j in range(6)) / scale
            worst = max(worst, err)
            assert err <= 2e-13, (seed, trial, kind, t, err, mu0, mu1, lh[t], T)
    assert n_stiff >= 40
    from parity import record
    record("pair_exponential_two_way_seed%d" % seed, worst_relative_to_norm=worst, stiff_intervals=int(n_stiff))
