#!/usr/bin/env python3
"""Gain ratio and gradient norm of every trust-region step of one lambda-correction solve, in float64 as the oracle (= the reference's
SciPy calls) evaluates them and in 60-digit arithmetic (mpmath): how far from SciPy's thresholds (radius doubled when the ratio
exceeds 0.75, stop when |J^T f| < 1e-10) the reference's own decisions sit.  CPU only.

    python tools/gain_ratio_study.py CANDIDATE [WORKLOAD [SOLVE]]      # default: config2, the solve with the most evaluations; SOLVE -1: config 5's interval 81
    python tools/gain_ratio_study.py 3427 config2 > profiles/rNN_gain_ratio_study.txt"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
warnings.simplefilter('ignore')
import numpy as np, mpmath as mp
mp.mp.dps = 50
from misti_amd import workloads
import oracle.misti_oracle as mo
from oracle.batch import oracle_eval, oracle_truth_spectrum
wl = sys.argv[2] if len(sys.argv) > 2 else 'config2'       # config2 | config5 | config5:default (the reference's default fit: the residual of CorrectLambda.py:94-110) ...
wl_name, _, wl_fit = wl.partition(':')
DEFAULT_FIT = wl_fit == 'default'
w = getattr(workloads, wl_name)(oracle_truth_spectrum, **({'cpfit': False} if DEFAULT_FIT else {}))
c = int(sys.argv[1]) if len(sys.argv) > 1 else 3427
which = int(sys.argv[3]) if len(sys.argv) > 3 else None
s, p = float(w.split_time[c]), list(w.params[c])
grab = []
class G(mo._PairChain):
    def solve_lambda_system(self, cpfit=True, prec=1e-10, norm_eps=0.02):
        self._n = 0
        r = super().solve_lambda_system(cpfit, prec, norm_eps)
        if self._n > 0: grab.append(dict(mu=self._mu_s, lh=self._lh_s, P0=[list(map(float, q)) for q in self.P0], evals=self._log))
        return r
    def _residual_cp(self, l):
        if self._n == 0:
            self._mu_s, self._lh_s, self._log = list(self.mu), list(self.lh), []
        self._n += 1
        r = super()._residual_cp(l)
        self._log.append((tuple(map(float, l)), tuple(map(float, r))))
        return r
    def _residual_ect(self, l):
        if self._n == 0:
            self._mu_s, self._lh_s, self._log = list(self.mu), list(self.lh), []
        self._n += 1
        r = super()._residual_ect(l)
        self._log.append((tuple(map(float, l)), tuple(map(float, r))))
        return r
old = mo._PairChain; mo._PairChain = G
try:
    oracle_eval(w.times, w.lh, w.bands, w.pulses, w.flags, w.sample_date, s, p, w.jsfs)
finally:
    mo._PairChain = old
g = grab[-1]                      # interval 84 is the last migrating solve of this candidate? check nfev
lens = [len(x['evals']) for x in grab]
k = int(np.argmax(lens)) if which is None else (which if which >= 0 else [i for i, x in enumerate(grab) if abs(x['lh'][0] - 0.07221297608789039) < 1e-12][0]); g = grab[k]
print('solve', k, 'of', len(grab), 'evals', lens[k], 'mu', g['mu'], 'lh', g['lh'])
mu0, mu1 = g['mu']; P0 = g['P0']; lh = g['lh']
def f_exact(x):
    M = mp.matrix([[-2*mu0 - x[0], 0, mu1], [0, -2*mu1 - x[1], mu0], [2*mu0, 2*mu1, -mu0 - mu1]])
    E = mp.expm(M)
    out = []
    if DEFAULT_FIT:
        # LambdaSystem (CorrectLambda.py:151-157,94-110) on the unit interval: conditional expected coalescence time of the pair chain minus the
        # one-population value 1 / lh - 1 / (e^lh - 1) (:74-77)
        Minv = M ** -1
        for kk in (0, 1):
            s_ = sum(mp.mpf(q) for q in P0[kk])
            pn = mp.matrix([mp.mpf(q) / s_ for q in P0[kk]])
            vec1 = Minv * (Minv * ((E - mp.eye(3)) * pn))
            vec2 = E * pn
            pnc = sum(vec2)
            vec2 = Minv * vec2
            ect2 = (x[0] * (vec2[0] - vec1[0]) + x[1] * (vec2[1] - vec1[1])) / (1 - pnc)
            lam = mp.mpf(lh[kk])
            out.append(ect2 - (1 / lam - 1 / (mp.e ** lam - 1)))
        return out
    for kk in (0, 1):
        v = E * mp.matrix(P0[kk])
        out.append(sum(v) - mp.e ** (-mp.mpf(lh[kk])) * sum(mp.mpf(q) for q in P0[kk]))
    return out
# reconstruct iterates: evaluations come in groups base, +h0, +h1
ev = g['evals']
bases = [(ev[i], ev[i+1], ev[i+2]) for i in range(0, len(ev) - 2, 3)]
def ratio_from(fb, fa, fc, xb, xa, xc, xn, fn):
    J = [[(fa[0]-fb[0])/(xa[0]-xb[0]), (fc[0]-fb[0])/(xc[1]-xb[1])], [(fa[1]-fb[1])/(xa[0]-xb[0]), (fc[1]-fb[1])/(xc[1]-xb[1])]]
    pstep = [xn[0]-xb[0], xn[1]-xb[1]]
    gk = [J[0][0]*fb[0] + J[1][0]*fb[1], J[0][1]*fb[0] + J[1][1]*fb[1]]
    Jp = [J[0][0]*pstep[0] + J[0][1]*pstep[1], J[1][0]*pstep[0] + J[1][1]*pstep[1]]
    pred = -(0.5*(Jp[0]**2 + Jp[1]**2) + pstep[0]*gk[0] + pstep[1]*gk[1])
    act = 0.5*(fb[0]**2+fb[1]**2) - 0.5*(fn[0]**2+fn[1]**2)
    return act/pred
for i in range(len(bases) - 1):
    (xb, fb), (xa, fa), (xc, fc) = bases[i]
    (xn, fn) = bases[i+1][0]
    r64 = ratio_from(fb, fa, fc, xb, xa, xc, xn, fn)
    fbe, fae, fce, fne = f_exact(xb), f_exact(xa), f_exact(xc), f_exact(xn)
    rex = ratio_from(fbe, fae, fce, [mp.mpf(v) for v in xb], [mp.mpf(v) for v in xa], [mp.mpf(v) for v in xc], [mp.mpf(v) for v in xn], fne)
    def gn(fb, fa, fc, xb, xa, xc):
        J = [[(fa[0]-fb[0])/(xa[0]-xb[0]), (fc[0]-fb[0])/(xc[1]-xb[1])], [(fa[1]-fb[1])/(xa[0]-xb[0]), (fc[1]-fb[1])/(xc[1]-xb[1])]]
        return max(abs(J[0][0]*fb[0] + J[1][0]*fb[1]), abs(J[0][1]*fb[0] + J[1][1]*fb[1]))
    g64 = gn(fb, fa, fc, xb, xa, xc); gex = gn(fbe, fae, fce, [mp.mpf(v) for v in xb], [mp.mpf(v) for v in xa], [mp.mpf(v) for v in xc])
    print('it', i, 'x (%.4f, %.4f)' % xb, 'ratio f64 %.6f exact %.6f' % (r64, float(rex)), ' g_norm f64 %.4e exact %.4e' % (g64, float(gex)))
# the point the solve stopped at (its gradient test fired, or the budget ended): the gradient norm alone, float64 against exact
(xb, fb), (xa, fa), (xc, fc) = bases[-1]
fbe, fae, fce = f_exact(xb), f_exact(xa), f_exact(xc)
def gn_last(fb, fa, fc, xb, xa, xc):
    J = [[(fa[0]-fb[0])/(xa[0]-xb[0]), (fc[0]-fb[0])/(xc[1]-xb[1])], [(fa[1]-fb[1])/(xa[0]-xb[0]), (fc[1]-fb[1])/(xc[1]-xb[1])]]
    return [J[0][0]*fb[0] + J[1][0]*fb[1], J[0][1]*fb[0] + J[1][1]*fb[1]]
g64 = gn_last(fb, fa, fc, xb, xa, xc); gex = gn_last(fbe, fae, fce, [mp.mpf(v) for v in xb], [mp.mpf(v) for v in xa], [mp.mpf(v) for v in xc])
print('last it', len(bases) - 1, 'x (%.4f, %.4f)' % xb, ' g f64 (%.4e, %.4e) exact (%.4e, %.4e)   [stop when max |g| < 1e-10]' % (g64[0], g64[1], float(gex[0]), float(gex[1])))
# MISTI_STUDY_POINT="x0,x1": the gradient an exact forward-difference Jacobian gives at ANOTHER point (e.g. the device's iterate of the same step,
# from tools/trace_candidate.py): is a different decision there a different ARITHMETIC or just a different POINT?
pt = os.environ.get("MISTI_STUDY_POINT")
if pt:
    x = tuple(float(v) for v in pt.split(","))
    h = [1.4901161193847656e-08 * max(1.0, abs(v)) for v in x]
    xa, xc = (x[0] + h[0], x[1]), (x[0], x[1] + h[1])
    ge = gn_last(f_exact(x), f_exact(xa), f_exact(xc), [mp.mpf(v) for v in x], [mp.mpf(v) for v in xa], [mp.mpf(v) for v in xc])
    print('at the point (%.16g, %.16g): exact g (%.4e, %.4e)' % (x[0], x[1], float(ge[0]), float(ge[1])))
