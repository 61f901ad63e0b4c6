"""CPU oracle for the MiSTI composite-likelihood hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``misti_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker.

This is a NumPy/SciPy *restatement* of the reference algorithm
(``/root/reference``; all ``file:line`` cites below are into that tree):

    MigrationInference.JAFSLikelihood   MigrationInference.py:566-614
      -> CorrectLambdas                 MigrationInference.py:305-378
           CorrectLambda.*              CorrectLambda.py:29-317
      -> Smooth / SmoothConst           MigrationInference.py:380-405
      -> JAFSpectrum / SolveDifEq       MigrationInference.py:467-540
           TwoPopulations.*             TwoPopulations.py:56-377
           OnePopulation.*              OnePopulation.py:37-178

Third-party arithmetic on the path lives in SciPy, which the reference does not
pin (no requirements file).  The oracle calls the same SciPy entry points at the
same call sites the reference does (``scipy.linalg.expm``, ``scipy.linalg.inv``,
``scipy.optimize.least_squares(method='trf')``, ``scipy.special.gammaln``); it is
pinned on SciPy 1.15.3 / NumPy 2.2.6 (the versions in this image) against
golden vectors produced by running the reference itself
(``tests/golden/make_golden.py`` -> ``tests/golden/*.json``).

Differences from the reference that do not change results:
  * the constant sparsity patterns (44x44 generator, state->JSFS table, pulse
    operator, ancient-sample map) are enumerated once at import instead of being
    re-derived by Python object manipulation on every interval;
  * the caller's ``times``/``lambdas`` lists are copied, not mutated, when the
    split time is fractional (MigrationInference.py:93-99 mutates them);
  * hard errors raise ``OracleError`` instead of ``sys.exit(0)``.
"""
from __future__ import annotations

import itertools
import math
from math import exp, log, sqrt

import numpy as np
from scipy import linalg, optimize, special

# Test hook (tools/self_perturbation.py --internal): a function applied to every pair-chain matrix exponential, to
# measure how far ONE ULP there moves the result - the reference's internal indeterminacy (tests/parity.py).
EXPM_HOOK = None

__all__ = ["OracleError", "OracleModel", "TWO_POP", "ONE_POP", "STATUS"]

STATUS = {"ok": 0, "negative_param": 1, "correction_failed": 2}


class OracleError(RuntimeError):
    """Raised where the reference prints a message and calls sys.exit(0)."""


# --------------------------------------------------------------------------
# State spaces (TwoPopulations.py:99-186, OnePopulation.py:64-107)
# --------------------------------------------------------------------------
# A lineage is (d0, d1, pop): number of sampled descendants in genome 1 and
# genome 2, and the population it currently sits in.  A state is a multiset of
# lineages with sum(d0) == sum(d1) == 2.

def _canon(state):
    # Ordering of CheckState (TwoPopulations.py:87-91): three stable sorts =>
    # primary d0+d1 descending, then d0 descending, then pop ascending.
    return tuple(sorted(state, key=lambda l: (-(l[0] + l[1]), -l[0], l[2])))


def _two_pop_state(ind):
    """Index -> state, TwoPopulations.py:130-186."""
    if ind < 9:
        a, b = ind // 3, ind % 3          # lineages of genome 1 / genome 2 in pop 1
        st = [(1, 0, 1 if k < a else 0) for k in range(2)]
        st += [(0, 1, 1 if k < b else 0) for k in range(2)]
    elif ind < 15:
        r = ind - 9
        st = [(2, 0, r // 3)] + [(0, 1, 1 if k < r % 3 else 0) for k in range(2)]
    elif ind < 23:
        r = ind - 15
        st = [(1, 1, r // 4), (1, 0, (r % 4) // 2), (0, 1, r % 2)]
    elif ind < 29:
        r = ind - 23
        st = [(0, 2, r // 3)] + [(1, 0, 1 if k < r % 3 else 0) for k in range(2)]
    elif ind < 33:
        r = ind - 29
        st = [(2, 1, r // 2), (0, 1, r % 2)]
    elif ind < 37:
        r = ind - 33
        st = [(1, 2, r // 2), (1, 0, r % 2)]
    elif ind < 41:
        r = ind - 37
        st = [(2, 0, r // 2), (0, 2, r % 2)]
    else:
        r = ind - 41                       # 41: both in 0, 42: one each, 43: both in 1
        st = [(1, 1, 1 if r == 2 else 0), (1, 1, 1 if r >= 1 else 0)]
    return _canon(st)


_JAF_CLASS = {(1, 0): 0, (2, 0): 1, (0, 1): 2, (1, 1): 3, (2, 1): 4, (0, 2): 5, (1, 2): 6}


class _TwoPopTables:
    """Constant structure of the 44-state chain (TwoPopulations.py)."""

    N = 44

    def __init__(self):
        N = self.N
        self.states = [_two_pop_state(i) for i in range(N)]
        self.index = {s: i for i, s in enumerate(self.states)}
        if len(self.index) != N:
            raise OracleError("two-population state codec is not a bijection")
        # Generator M = la0*A[0] + la1*A[1] + mu0*B[0] + mu1*B[1]; column = source
        # state (UpdateMatrixCol, TwoPopulations.py:336-359).
        # ``events[src]`` keeps the reference's order of additions so that the
        # assembled matrix is bit-identical to the reference's.
        self.A = np.zeros((2, N, N))
        self.B = np.zeros((2, N, N))
        self.events = []
        for src, st in enumerate(self.states):
            ev = []
            for i, (d0, d1, p) in enumerate(st):
                moved = list(st)
                moved[i] = (d0, d1, 1 - p)
                dst = self.index[_canon(moved)]
                self.B[p][dst, src] += 1.0
                self.B[p][src, src] -= 1.0
                ev.append((dst, 2 + p))
                for j in range(i + 1, len(st)):
                    e0, e1, q = st[j]
                    if q != p:
                        continue
                    rest = [l for k, l in enumerate(st) if k not in (i, j)]
                    rest.append((d0 + e0, d1 + e1, p))
                    if len(rest) >= 2:          # 2 -> 1 is absorption (:356)
                        dst = self.index[_canon(rest)]
                        self.A[p][dst, src] += 1.0
                    else:
                        dst = -1
                    self.A[p][src, src] -= 1.0
                    ev.append((dst, p))
            self.events.append(ev)
        # state -> JSFS class counts (StateToJAF, TwoPopulations.py:188-219)
        self.jaf = np.zeros((N, 7))
        for i, st in enumerate(self.states):
            for d0, d1, _ in st:
                self.jaf[i, _JAF_CLASS[(d0, d1)]] += 1.0
        # States that never change without migration: two lineages in different
        # populations (TwoPopulations.py:67-71).
        self.stationary = [i for i, st in enumerate(self.states)
                           if len(st) == 2 and st[0][2] != st[1][2]]
        # Signature used to give mass back to the deleted states
        # (TwoPopulations.py:273-283): descendants located in population 1.
        self.signature = [(sum(l[0] * l[2] for l in st), sum(l[1] * l[2] for l in st))
                          for st in self.states]
        # AncientSampleP0 (TwoPopulations.py:246-262)
        self.ancient = np.zeros((N, N))
        for i, st in enumerate(self.states):
            if sum(1 for l in st if l == (1, 0, 0)) == 2:
                self.ancient[2, i] += 1.0
            if sum(1 for l in st if l == (2, 0, 0)) == 1:
                self.ancient[11, i] += 1.0
        # PulseMigration (TwoPopulations.py:361-377): every lineage in pop1 moves
        # independently with probability r.  entries: (dst, src, n_stay, n_move)
        self.pulse = {0: [], 1: []}
        for pop1 in (0, 1):
            for src, st in enumerate(self.states):
                movers = [k for k, l in enumerate(st) if l[2] == pop1]
                for choice in itertools.product((0, 1), repeat=len(movers)):
                    new = list(st)
                    for k, c in zip(movers, choice):
                        if c:
                            new[k] = (st[k][0], st[k][1], 1 - pop1)
                    dst = self.index[_canon(new)]
                    self.pulse[pop1].append((dst, src, len(movers) - sum(choice), sum(choice)))
        # CollapsePops ranges (MigrationInference.py:518-528)
        self.collapse = [(0, 9), (9, 15), (15, 23), (23, 29), (29, 33), (33, 37), (37, 41), (41, 44)]

    def generator(self, la, mu):
        """Dense generator, additions in the order of UpdateMatrixCol (:336-359)."""
        rate = (la[0], la[1], mu[0], mu[1])
        M = np.zeros((self.N, self.N))
        for src, ev in enumerate(self.events):
            total = 0.0
            for dst, kind in ev:
                if dst >= 0:
                    M[dst, src] += rate[kind]
                total += rate[kind]
            M[src, src] -= total
        return M

    def pulse_matrix(self, r, pop1):
        P = np.zeros((self.N, self.N))
        for dst, src, ns, nm in self.pulse[pop1]:
            P[dst, src] += (1.0 - r) ** ns * r ** nm
        return P


class _OnePopTables:
    """Constant structure of the 8-state chain (OnePopulation.py)."""

    N = 8

    def __init__(self):
        raw = [
            [(1, 0), (1, 0), (0, 1), (0, 1)],
            [(2, 0), (0, 1), (0, 1)],
            [(1, 1), (1, 0), (0, 1)],
            [(0, 2), (1, 0), (1, 0)],
            [(2, 1), (0, 1)],
            [(1, 2), (1, 0)],
            [(2, 0), (0, 2)],
            [(1, 1), (1, 1)],
        ]                                                   # OnePopulation.py:86-107
        canon = lambda s: tuple(sorted(s, key=lambda l: (-(l[0] + l[1]), -l[0])))
        self.states = [canon(s) for s in raw]
        self.index = {s: i for i, s in enumerate(self.states)}
        self.A = np.zeros((8, 8))                           # generator / la
        self.events = []                                    # reference order of additions
        for src, st in enumerate(self.states):              # OnePopulation.py:160-178
            ev = []
            for i in range(len(st)):
                for j in range(i + 1, len(st)):
                    rest = [l for k, l in enumerate(st) if k not in (i, j)]
                    rest.append((st[i][0] + st[j][0], st[i][1] + st[j][1]))
                    dst = -1
                    if len(rest) >= 2:
                        dst = self.index[canon(rest)]
                        self.A[dst, src] += 1.0
                    self.A[src, src] -= 1.0
                    ev.append(dst)
            self.events.append(ev)
        self.jaf = np.zeros((8, 7))                         # OnePopulation.py:109-136
        for i, st in enumerate(self.states):
            for d0, d1 in st:
                self.jaf[i, _JAF_CLASS[(d0, d1)]] += 1.0

    def generator(self, la):
        """Dense generator, additions in the order of UpdateMatrixCol (:160-178)."""
        M = np.zeros((8, 8))
        for src, ev in enumerate(self.events):
            total = 0.0
            for dst in ev:
                if dst >= 0:
                    M[dst, src] += la
                total += la
            M[src, src] -= total
        return M


TWO_POP = _TwoPopTables()
ONE_POP = _OnePopTables()


# --------------------------------------------------------------------------
# lambda-correction: the 3-state chain of one genome's lineage pair
# (CorrectLambda.py).  States: both in pop 0, both in pop 1, one in each.
# --------------------------------------------------------------------------
class _PairChain:
    def __init__(self, mixture_th=0.0):
        self.mixtureTH = mixture_th                         # CorrectLambda.py:31,44-45
        self.count_solver_calls = 0
        self.count_fun_evals = 0

    def set_mu(self, mu0, mu1):
        self.mu = [mu0, mu1]

    def set_interval(self, lh, T, P0):                      # CorrectLambda.py:50-53
        self.lh = [lh[0], lh[1]]
        self.T = T
        self.P0 = P0

    def _matrix(self, l):                                   # CorrectLambda.py:55-56
        mu = self.mu
        return np.array([[-2 * mu[0] - l[0], 0.0, mu[1]],
                         [0.0, -2 * mu[1] - l[1], mu[0]],
                         [2 * mu[0], 2 * mu[1], -mu[0] - mu[1]]])

    def _expm(self, M):                                     # CorrectLambda.py:58-62
        self.count_fun_evals += 1
        e = linalg.expm(np.dot(M, self.T))
        return e if EXPM_HOOK is None else EXPM_HOOK(e)

    # -- one-population helpers ------------------------------------------
    def ect_one_pop(self, lam):                             # CorrectLambda.py:67-72
        r = 0 if lam > 100 else self.T / (exp(lam * self.T) - 1)
        return 1.0 / lam - r

    def ect_one_pop_tmp(self, lam):                         # CorrectLambda.py:74-77
        pnc = exp(-lam * self.T)
        return 1.0 / lam - self.T / (1.0 / pnc - 1.0), pnc

    def ect_one_pop_noncond(self, lam):                     # CorrectLambda.py:79-80
        return (1 - exp(-lam * self.T) * (1 + lam * self.T)) / lam

    def fit_single_pop(self):                               # CorrectLambda.py:82-92
        pnc = [sum(self.P0[0]), sum(self.P0[1])]
        pnc = [pnc[0] / sum(pnc), pnc[1] / sum(pnc)]
        Te = pnc[0] * self.ect_one_pop(self.lh[0]) + pnc[1] * self.ect_one_pop(self.lh[1])
        x0 = pnc[0] * self.lh[0] + pnc[1] * self.lh[1]
        lower = 0.01 * min(self.lh[0], self.lh[1])
        self.count_solver_calls += 1
        res = optimize.least_squares(lambda lam: self.ect_one_pop(lam[0]) - Te, x0,
                                     bounds=(lower, np.inf), gtol=1e-10, xtol=1e-10)
        return res.x

    # -- two-population residuals ---------------------------------------
    def _residual_cp(self, l):                              # LambdaSystem1, :169-173,135-144
        MET = self._expm(self._matrix(l))
        out = []
        for k in (0, 1):
            nch = exp(-self.lh[k] * self.T) * sum(self.P0[k])
            out.append(sum(np.dot(MET, self.P0[k])) - nch)
        return np.array(out)

    def _residual_ect(self, l):                             # LambdaSystem, :151-157,94-110
        M = self._matrix(l)
        MET = self._expm(M)
        Minv = linalg.inv(M)
        res = []
        for k in (0, 1):
            pn = [v / sum(self.P0[k]) for v in self.P0[k]]
            vec1 = np.dot(MET - np.identity(3), pn)
            vec1 = np.dot(Minv, np.dot(Minv, vec1))
            vec2 = np.dot(MET, pn)
            pnc = sum(vec2)
            vec2 = np.dot(self.T, np.dot(Minv, vec2))
            vec = vec2 - vec1
            ect2 = (l[0] * vec[0] + l[1] * vec[1]) / (1 - pnc)
            res.append(ect2 - self.ect_one_pop_tmp(self.lh[k])[0])
        return (res[0], res[1])

    def _residual_nomig(self, l):                           # LambdaSystemNoMigration, :237-251
        out = []
        for k in (0, 1):
            pr = self.pr0[k]
            pnc = pr[0] * exp(-l[0] * self.T) + pr[1] * exp(-l[1] * self.T) + pr[2]
            ct = (pr[0] * self.ect_one_pop_noncond(l[0]) + pr[1] * self.ect_one_pop_noncond(l[1])) / (1 - pnc)
            out.append(ct - self.ect_one_pop(self.lh[k]))
        return (out[0], out[1])

    def _decay(self, lc):                                   # CorrectLambda.py:233-234,262-263
        T, P0 = self.T, self.P0
        return [[P0[k][0] * exp(-lc[0] * T), P0[k][1] * exp(-lc[1] * T), P0[k][2]] for k in (0, 1)]

    def solve_no_migration_cp(self):                        # SolveNoMigration1, :213-235
        P0, T = self.P0, self.T
        s0, s1 = sum(P0[0]), sum(P0[1])
        A1, A2, A3, A4 = P0[0][0] / s0, P0[0][1] / s0, P0[1][0] / s1, P0[1][1] / s1
        C1, C2 = P0[0][2] / s0, P0[1][2] / s1
        D = A1 * A4 - A2 * A3
        B1, B2, B3, B4 = A4 / D, -A2 / D, -A3 / D, A1 / D
        X1 = exp(-self.lh[0] * T) - C1
        X2 = exp(-self.lh[1] * T) - C2
        if B1 * X1 + B2 * X2 > 0 and B3 * X1 + B4 * X2 > 0:
            lc = [-log(B1 * X1 + B2 * X2) / T, -log(B3 * X1 + B4 * X2) / T]
        else:
            lc = [-1, -1]
        return [lc, self._decay(lc)]

    def solve_no_migration(self):                           # SolveNoMigration, :253-264
        self.pr0 = [[v / sum(self.P0[k]) for v in self.P0[k]] for k in (0, 1)]
        lower = 0.01 * min(self.lh[0], self.lh[1])
        self.count_solver_calls += 1
        res = optimize.least_squares(self._residual_nomig, self.lh, bounds=(lower, np.inf),
                                     gtol=1e-10, xtol=1e-10)
        lc = res.x
        return [lc, self._decay(lc)]

    def solve_lambda_system(self, cpfit=True, prec=1e-10, norm_eps=0.02):   # :266-317
        P0 = self.P0
        s0, s1 = sum(P0[0]), sum(P0[1])
        mixture = sqrt(sum((P0[0][i] / s0 - P0[1][i] / s1) ** 2 for i in range(3)))
        if mixture < self.mixtureTH:
            return [[-1, -1], P0]
        if self.mu[0] + self.mu[1] < prec:
            return self.solve_no_migration_cp() if cpfit else self.solve_no_migration()
        n0 = sqrt(sum(P0[0][i] ** 2 for i in range(3)))
        n1 = sqrt(sum(P0[1][i] ** 2 for i in range(3)))
        nd = sqrt(sum((P0[0][i] - P0[1][i]) ** 2 for i in range(3)))
        if nd < norm_eps * min(n0, n1):
            mean = (self.lh[0] + self.lh[1]) / 2.0
            self.lh = [mean, mean]
        # stretch the interval to unit length (:293-298)
        T_keep = self.T
        self.T = self.T / T_keep
        self.mu = [self.mu[0] * T_keep, self.mu[1] * T_keep]
        self.lh = [self.lh[0] * T_keep, self.lh[1] * T_keep]
        fun = self._residual_cp if cpfit else self._residual_ect
        self.count_solver_calls += 1
        res = optimize.least_squares(fun, [self.lh[0], self.lh[1]], bounds=(-np.inf, np.inf),
                                     gtol=prec, xtol=prec)
        x = res.x
        self.T = T_keep
        self.mu = [self.mu[0] / T_keep, self.mu[1] / T_keep]
        self.lh = [self.lh[0] / T_keep, self.lh[1] / T_keep]
        l = [x[0] / T_keep, x[1] / T_keep]
        MET = self._expm(self._matrix(l))
        return [l, [np.dot(MET, P0[0]), np.dot(MET, P0[1])]]

    def coal_rates(self, l):                                # CoalRates, :112-122 (forward map)
        MET = self._expm(self._matrix(l))
        p0, lh = [None, None], [None, None]
        for k in (0, 1):
            p0[k] = np.dot(MET, self.P0[k])
            lh[k] = -log(sum(p0[k]) / sum(self.P0[k])) / self.T
        return lh, p0


def _pulse_pairs(p0, pu):
    """Pulse migration on the two pair-state vectors, MigrationInference.py:315-323."""
    rate = pu[0] + pu[1]
    if not rate > 0:
        return p0
    a = 0 if pu[0] > 0 else 1
    b = 1 - a
    out = []
    for k in (0, 1):
        n = [None, None, None]
        n[a] = p0[k][a] * (1 - rate) ** 2
        n[b] = p0[k][a] * rate ** 2 + p0[k][b] + p0[k][2] * rate
        n[2] = p0[k][a] * 2 * (1 - rate) * rate + p0[k][2] * (1 - rate)
        out.append(n)
    return out


# --------------------------------------------------------------------------
# The engine (MigrationInference.py)
# --------------------------------------------------------------------------
class OracleModel:
    """Restatement of class MigrationInference (MigrationInference.py:35-739).

    Constructor arguments and keyword flags have the reference's meaning:
    ``OracleModel(times, lambdas, dataJAFS, splitT, mi, pu, smooth=, cpfit=,
    trueEPS=, unfolded=, sampleDate=, mixtureTH=)``.
    """

    def __init__(self, times, lambdas, dataJAFS, splitT, mi=(), pu=(), **kw):
        self.cpfit = bool(kw.get("cpfit", False))
        self.correct = not bool(kw.get("trueEPS", False))
        self.smooth = bool(kw.get("smooth", False))
        self.unfolded = bool(kw.get("unfolded", False))
        self.sampleDate = kw.get("sampleDate", 0)
        if splitT < self.sampleDate:                                    # :85-86
            raise OracleError("split time more recent than sample date")
        times = [float(t) for t in times]
        lambdas = [[float(l[0]), float(l[1])] for l in lambdas]
        frac = splitT % 1                                               # :89-99
        splitT = int(splitT)
        if splitT - 1 > len(times):
            raise OracleError("invalid split time")
        if frac != 0.0:
            t1 = frac * times[splitT]
            t2 = times[splitT] - t1
            times[splitT] = t1
            times.insert(splitT + 1, t2)
            lambdas.insert(splitT + 1, list(lambdas[splitT]))
            splitT += 1
        self.lh = lambdas
        self.times = times
        self.numT = len(self.lh)
        if len(self.times) != self.numT - 1:                            # :105-107
            raise OracleError("unexpected number of time intervals")
        self.splitT = splitT
        self.set_model(mi, pu)
        self.set_jafs(dataJAFS)
        self.lc = [[1, 1] for _ in range(self.numT)]
        self.cl = _PairChain(kw.get("mixtureTH", 0.0))
        self.JAFS = None
        self.Pr = None
        self.llh = None
        self.status = 0

    # -- model -----------------------------------------------------------
    def set_model(self, mis, pus):                                      # :229-289
        n = self.numT
        self.mi = [[None, None] for _ in range(n)]
        self.pu = [[None, None] for _ in range(n)]
        self.optMis, self.optPus = [], []
        for el in mis:
            pop = int(el[0]) - 1
            if pop not in (0, 1):
                raise OracleError("population index should be 1 or 2")
            start, end = int(el[1]), int(el[2])
            if start < self.sampleDate:
                raise OracleError("migration start before sample date")
            if end <= start:
                raise OracleError("migration start should be strictly less than end")
            val, opt = float(el[3]), int(el[4])
            for i in range(start, end):
                if self.mi[i][pop] is not None:
                    raise OracleError("migration rate intervals should not overlap")
                self.mi[i][pop] = val
            if opt == 1:
                self.optMis.append([pop, start, end, val])
        for el in pus:
            pop = int(el[0]) - 1
            if pop not in (0, 1):
                raise OracleError("population index should be 1 or 2")
            t = int(el[1])
            if t < self.sampleDate:
                raise OracleError("pulse time before sample date")
            val, opt = float(el[2]), int(el[3])
            if val < 0 or val > 1:
                raise OracleError("pulse migration rate should be between 0 and 1")
            if self.pu[t][0] is not None or self.pu[t][1] is not None:
                raise OracleError("only single-direction pulse migration at a time")
            self.pu[t][pop] = val
            if opt == 1:
                self.optPus.append([pop, t, val])
        for arr in (self.mi, self.pu):
            for row in arr:
                for k in (0, 1):
                    if row[k] is None:
                        row[k] = 0.0

    def map_parameters(self, params):                                   # :291-298
        if len(params) != len(self.optMis) + len(self.optPus):
            raise OracleError("incorrect number of parameters")
        for i, (pop, start, end, _) in enumerate(self.optMis):
            for j in range(start, end):
                self.mi[j][pop] = params[i]
        for i, (pop, t, _) in enumerate(self.optPus):
            self.pu[t][pop] = params[len(self.optMis) + i]

    def set_jafs(self, dataJAFS):                                       # :202-227
        if len(dataJAFS) != 8:
            raise OracleError("unexpected data SFS")
        self.dataJAFS = [float(v) for v in dataJAFS[1:]]
        d = self.dataJAFS
        self.snps = sum(d)
        g = special.gammaln
        c = g(self.snps + 1)
        if self.unfolded:
            for i in range(7):
                c -= g(d[i] + 1)
        else:
            c -= g(d[0] + d[6] + 1)
            c -= g(d[1] + d[5] + 1)
            c -= g(d[2] + d[4] + 1)
            c -= g(d[3] + 1)
        self.llh_const = float(c)

    # -- lambda correction -----------------------------------------------
    def correct_lambdas(self):                                          # :305-378
        p0 = [[1, 0, 0], [0, 1, 0]]
        self.Pr = [[[1.0, 0.0], [0.0, 1.0], [0.0, 0.0]]]
        nc = [0, 0]
        cl = self.cl
        # Diagnostic, not part of the reference: largest corrected rate x interval length before
        # smoothing.  Beyond ~30 the pair has coalesced to 1e-13 inside the interval, the
        # correction's residual is flat in that rate and SciPy's solver stops where rounding noise
        # in its finite-difference Jacobian lets the gradient test pass ("runaway" rate).
        self.max_rate_x_len = 0.0
        for t in range(self.splitT):
            p0 = _pulse_pairs(p0, self.pu[t])
            cl.set_mu(self.mi[t][0], self.mi[t][1])
            if not self.correct:
                self.lc[t] = [self.lh[t][0], self.lh[t][1]]
            else:
                cl.set_interval(self.lh[t], self.times[t], p0)
                sol = cl.solve_lambda_system(self.cpfit)
                self.lc[t] = [sol[0][0], sol[0][1]]
                if sol[0][0] <= 0 or sol[0][1] <= 0:
                    return False
                self.max_rate_x_len = max(self.max_rate_x_len, sol[0][0] * self.times[t], sol[0][1] * self.times[t])
                p0 = sol[1]
            self.Pr.append([[p0[0][0], p0[1][0]], [p0[0][1], p0[1][1]], [p0[0][2], p0[1][2]]])
            nc = [sum(p0[0]), sum(p0[1])]   # a probability here, used as a log below (:353-354)
        for t in range(self.splitT, self.numT - 1):
            T = self.times[t]
            if T == 0:
                self.lc[t] = [1, 1]
                continue
            if not self.cpfit:
                cl.set_interval(self.lh[t], T, [[exp(nc[0]), 0, 0], [exp(nc[1]), 0, 0]])
                lam = cl.fit_single_pop()[0]
            else:
                pnc = (exp(-T * self.lh[t][0]) + exp(nc[1] - nc[0] - T * self.lh[t][1])) / (1 + exp(nc[1] - nc[0]))
                lam = -log(pnc) / T
            self.lc[t] = [lam, lam]
            nc = [nc[0] - T * lam, nc[1] - T * lam]
        t = self.numT - 1
        pr0, pr1 = exp(nc[0]), exp(nc[1])
        lam = (pr0 + pr1) / (pr0 / self.lh[t][0] + pr1 / self.lh[t][1])
        self.lc[t] = [lam, lam]
        if self.smooth:                                                 # :380-385
            self._smooth_const(0)
            self._smooth_const(1)
        return True

    def _smooth_const(self, k):                                         # :387-405
        i = 0
        lam = self.lh[0][k]
        while i < self.splitT:
            j = i
            acc, tsum = 0.0, 0.0
            while abs(self.lh[j][k] - lam) < 1e-10 and j < self.numT - 1:
                acc += self.lc[j][k] * self.times[j]
                tsum += self.times[j]
                j += 1
                if j == self.splitT:
                    break
            for m in range(i, j):
                self.lc[m][k] = acc / tsum
            lam = self.lh[j][k]
            i = j

    # -- expected spectrum -----------------------------------------------
    def jaf_spectrum(self):                                             # :467-506
        tp, op = TWO_POP, ONE_POP
        jafs = np.zeros(7)
        P0 = np.zeros(tp.N)
        P0[2] = 1.0
        for it in range(self.numT):
            two = it < self.splitT
            if two:
                mu = self.mi[it]
                if it == self.numT - 1 and mu[0] + mu[1] == 0.0:
                    raise OracleError("infinite coalescent time, no migration")
                la = self.lc[it]
                if mu[0] < 0 or mu[1] < 0 or la[0] < 0 or la[1] < 0:    # TwoPopulations.py:58-61
                    raise OracleError("negative rate")
            if it == self.sampleDate:
                P0 = tp.ancient.dot(P0)
            pu = self.pu[it][0] + self.pu[it][1]
            if two and pu > 0:
                P0 = tp.pulse_matrix(pu, 0 if self.pu[it][0] > 0 else 1).dot(P0)
            if it == self.splitT:
                P0 = np.array([sum(P0[a:b]) for a, b in tp.collapse])
            if two:
                M = tp.generator(la, mu)
                jaf = tp.jaf
                deleted = tp.stationary if mu[0] + mu[1] == 0 else []
            else:
                M = op.generator(self.lc[it][0])
                jaf = op.jaf
                deleted = []
            full0 = P0
            if deleted:                                                 # TwoPopulations.py:231-244
                keep = [i for i in range(tp.N) if i not in deleted]
                M = M[np.ix_(keep, keep)]
                P0 = P0[keep]
            if it < self.numT - 1:                                      # SolveDifEq :530-540
                T = self.times[it]
                P1 = np.dot(linalg.expm(np.dot(M, T)), P0)
            else:
                P1 = np.zeros(len(P0))
            integ = np.dot(linalg.inv(M), P1 - P0)
            if deleted:                                                 # TwoPopulations.py:264-309
                P1f = np.zeros(tp.N)
                If = np.zeros(tp.N)
                P1f[keep] = P1
                If[keep] = integ
                nP, nI = P1f.copy(), If.copy()
                for ind in deleted:
                    for i in range(tp.N):
                        if tp.signature[i] == tp.signature[ind]:
                            nP[ind] += full0[i] - P1f[i]
                            nI[ind] += T * full0[i] - If[i]
                P1, integ = nP, nI
            P0 = P1
            w = jaf if it >= self.sampleDate else jaf * np.array([1, 1, 0, 0, 0, 0, 0.0])
            for i in range(len(integ)):          # same order of additions as :501-506
                jafs = jafs + w[i] * integ[i]
        self.JAFS = [float(v) for v in jafs]

    # -- likelihood --------------------------------------------------------
    def jafs_likelihood(self, mu=()):                                   # :566-614
        self.llh = -10 ** 9
        self.status = 0
        for v in mu:
            if v < 0:
                self.status = STATUS["negative_param"]
                return -np.inf
        self.map_parameters(mu)
        if not self.correct_lambdas():
            self.status = STATUS["correction_failed"]
            return -np.inf
        self.jaf_spectrum()
        norm = sum(self.JAFS)
        self.JAFS = [v / norm for v in self.JAFS]
        self.llh = self.llk_for(self.dataJAFS, self.llh_const)
        return self.llh

    def llk_for(self, d, llh_const):                                    # :600-609
        J = self.JAFS
        llh = llh_const
        if not self.unfolded:
            llh += (d[0] + d[6]) * log(J[0] + J[6])
            llh += (d[1] + d[5]) * log(J[1] + J[5])
            llh += (d[2] + d[4]) * log(J[2] + J[4])
            llh += d[3] * log(J[3])
        else:
            for i in range(7):
                llh += d[i] * log(J[i])
        return llh

    # -- forward map (TestModel route) -------------------------------------
    def coalescent_rates(self, hold_mu=False):                          # :542-564
        """True rates ``lh`` + migration -> PSMC-like rates; overwrites self.lh.

        The reference never calls SetMu here: its CorrectLambda object keeps the migration rates
        of the LAST two-population interval from the preceding CorrectLambdas loop (:324) and
        applies them to every interval.  ``hold_mu=True`` restates exactly that; the default
        applies each interval's own rates (the model as specified)."""
        self.Pr = []
        for i in range(self.numT):
            self.lc[i] = [self.lh[i][0], self.lh[i][1]]
        p0 = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]
        for t in range(self.splitT):
            p0 = _pulse_pairs(p0, self.pu[t])
            if t == 0:
                self.Pr.append([[p0[0][0], p0[1][0]], [p0[0][1], p0[1][1]], [p0[0][2], p0[1][2]]])
            tm = self.splitT - 1 if hold_mu else t
            self.cl.set_mu(self.mi[tm][0], self.mi[tm][1])
            self.cl.set_interval(self.lh[t], self.times[t], p0)
            self.lh[t], p0 = self.cl.coal_rates(self.lc[t])
            self.Pr.append([[p0[0][0], p0[1][0]], [p0[0][1], p0[1][1]], [p0[0][2], p0[1][2]]])
        return self.lh


def llh_const(row, unfolded):
    """llh_const of MigrationInference.SetJAFS (:217-227) for one JSFS row of 8."""
    d = [float(v) for v in row[1:]]
    g = special.gammaln
    c = g(sum(d) + 1)
    if unfolded:
        for v in d:
            c -= g(v + 1)
    else:
        c -= g(d[0] + d[6] + 1)
        c -= g(d[1] + d[5] + 1)
        c -= g(d[2] + d[4] + 1)
        c -= g(d[3] + 1)
    return float(c)
