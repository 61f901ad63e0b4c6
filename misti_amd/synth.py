"""Synthetic two-population inputs for tests and benchmarks.

No simulator (ms / msHOT-lite / psmc) exists in this environment, so inputs are
synthesised directly in the file formats MiSTI reads:

  * ``psmc_text``   - a PSMC output with the program's own time grid
                      t_j = 0.1*(exp(j/(n-1)*ln(1+10*15)) - 1) and piecewise-constant
                      relative sizes drawn log-uniformly from a seeded RNG;
  * ``jsfs_text``   - a ``#MiSTI_JSFS version 1.0`` table;
  * ``self_consistent`` - a pair of PSMC-like rate trajectories derived from a
                      *true* two-population model (true rates + migration band +
                      pulse) with the forward map of the pair chain, so that the
                      lambda-correction has a solution near the truth.  With purely
                      random rates the correction fails for most candidates and
                      early exits inflate throughput.

All of this is host-side data preparation, not the hot path.
"""
from __future__ import annotations

import io as _io
import math
import random

import numpy as np

from . import io as mio

THETA_1 = 0.0521          # must differ between the two files, else the merged grid
THETA_2 = 0.0467          # has zero-length intervals


def psmc_grid(n):
    return [0.1 * (math.exp(j / (n - 1) * math.log(1 + 10 * 15)) - 1) for j in range(n)]


def psmc_text(n, seed, theta, rho=0.0093, lo=0.5, hi=4.0, run=2, rounds=1):
    """Text of a PSMC output with ``n`` RS rows; sizes constant in runs of ``run``."""
    rng = random.Random(seed)
    grid = psmc_grid(n)
    out = []
    for rd in range(rounds):
        lam = []
        while len(lam) < n:
            v = math.exp(rng.uniform(math.log(lo), math.log(hi)))
            lam += [v] * run
        lam = lam[:n]
        out.append("RD\t%d" % rd)
        out.append("LK\t-1000.0")
        out.append("QD\t0.0 -> 0.0")
        out.append("RI\t0.01")
        out.append("TR\t%.6f\t%.6f" % (theta, rho))
        out.append("MT\t15.0")
        for k in range(n):
            out.append("RS\t%d\t%.6f\t%.6f\t0.001\t0.001\t0.001" % (k, grid[k], lam[k]))
        out.append("PA\t4+25*2+4+6 %.6f %.6f 15.0" % (theta, rho))
        out.append("//")
    return "\n".join(out) + "\n"


def psmc_pair(n1, n2, seeds=(1, 2), sample_date=0.0, units=None, **kw):
    """Merged InputData of two synthetic PSMC files (reads its own text back)."""
    d1 = mio.read_psmc_file(_io.StringIO(psmc_text(n1, seeds[0], THETA_1, **kw)))
    d2 = mio.read_psmc_file(_io.StringIO(psmc_text(n2, seeds[1], THETA_2, **kw)))
    return mio.merge_psmc(d1, d2, sample_date, units)


def jsfs_text(rows):
    return mio.format_jsfs(rows)


# -- forward map of the pair chain ------------------------------------------
def _expm3(A):
    """exp of a small dense matrix: scaling and squaring with a Taylor kernel."""
    A = np.asarray(A, dtype=float)
    nrm = np.abs(A).sum(axis=0).max()
    s = max(0, int(math.ceil(math.log2(max(nrm, 1e-300) / 0.25)))) if nrm > 0.25 else 0
    B = A / (2.0 ** s)
    E = np.eye(A.shape[0])
    term = np.eye(A.shape[0])
    for k in range(1, 20):
        term = term.dot(B) / k
        E = E + term
    for _ in range(s):
        E = E.dot(E)
    return E


def pair_generator(l, mu):
    """3-state chain of one genome's two lineages: (both in 0, both in 1, one each)."""
    return np.array([[-2 * mu[0] - l[0], 0.0, mu[1]],
                     [0.0, -2 * mu[1] - l[1], mu[0]],
                     [2 * mu[0], 2 * mu[1], -mu[0] - mu[1]]])


def pulse_pairs(p0, pu):
    rate = pu[0] + pu[1]
    if not rate > 0:
        return p0
    a = 0 if pu[0] > 0 else 1
    b = 1 - a
    out = np.empty_like(p0)
    for k in (0, 1):
        out[k, a] = p0[k, a] * (1 - rate) ** 2
        out[k, b] = p0[k, a] * rate ** 2 + p0[k, b] + p0[k, 2] * rate
        out[k, 2] = p0[k, a] * 2 * (1 - rate) * rate + p0[k, 2] * (1 - rate)
    return out


def forward_rates(times, lc, split, mi, pu):
    """True per-population rates + migration -> the rates PSMC would see.

    For t < split the PSMC-like rate of genome k is -log(P[no coalescence in the
    interval])/T under the pair chain started from that genome's pair
    distribution (the quantity CorrectLambda.CoalRates computes in the
    reference); for t >= split the population is single and lh = lc.
    """
    lh = [list(map(float, r)) for r in lc]
    p0 = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    for t in range(split):
        p0 = pulse_pairs(p0, pu[t])
        E = _expm3(pair_generator(lc[t], mi[t]) * times[t])
        for k in (0, 1):
            p1 = E.dot(p0[k])
            lh[t][k] = -math.log(p1.sum() / p0[k].sum()) / times[t]
            p0[k] = p1
    return lh


def true_model(inp, split, seed=7, lo=0.6, hi=1.8, run=4):
    """A smooth-ish random truth on the merged grid of ``inp``: rates constant in
    runs of ``run`` intervals before the split, one shared trajectory after it."""
    rng = random.Random(seed)
    n = len(inp.lambdas)
    lc = []
    cur = [1.0, 1.0]
    for t in range(n):
        if t % run == 0:
            v0 = math.exp(rng.uniform(math.log(lo), math.log(hi)))
            v1 = math.exp(rng.uniform(math.log(lo), math.log(hi)))
            cur = [v0, v1] if t < split else [v0, v0]
        if t >= split:
            cur = [cur[0], cur[0]]
        lc.append(list(cur))
    return lc


def expand_model(n, mis, pus):
    """-mi / -pu descriptors -> per-interval arrays (MigrationInference.SetModel semantics)."""
    mi = [[0.0, 0.0] for _ in range(n)]
    pu = [[0.0, 0.0] for _ in range(n)]
    for pop, start, end, val, _opt in mis:
        for t in range(int(start), int(end)):
            mi[t][int(pop) - 1] = float(val)
    for pop, t, val, _opt in pus:
        pu[int(t)][int(pop) - 1] = float(val)
    return mi, pu


def self_consistent(inp, split, mis=(), pus=(), seed=7, **kw):
    """(times, lh, lc_true) with lh derived from a random truth by the forward map."""
    n = len(inp.lambdas)
    lc = true_model(inp, split, seed, **kw)
    mi, pu = expand_model(n, mis, pus)
    lh = forward_rates(inp.times, lc, split, mi, pu)
    return list(inp.times), lh, lc


def counts_from_spectrum(jafs, n_sites=10 ** 6, total=None):
    """Expected spectrum -> one JSFS row [total, 7 classes] of rounded counts."""
    s = sum(jafs)
    cls = [float(round(n_sites * v / s)) for v in jafs]
    return [float(total if total is not None else 30 * n_sites)] + cls


def chunk_rows(row, n_chunks=20, seed=11):
    """Split one JSFS row into ``n_chunks`` chunk rows (multinomially) for bootstrapping."""
    rng = np.random.default_rng(seed)
    p = np.full(n_chunks, 1.0 / n_chunks)
    parts = [rng.multinomial(int(v), p) for v in row]
    return [[float(parts[c][k]) for c in range(8)] for k in range(n_chunks)]
