"""Candidate 6761 of config3 under the default fit, interval 12: the reference-structured residual and its forward-difference Jacobian
against the exact ones (mpmath): which Gauss-Newton step is the biased one?"""
import sys, warnings
import numpy as np, mpmath as mp
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.simplefilter('ignore')
import oracle.misti_oracle as mo
from misti_amd import workloads
from oracle.batch import oracle_truth_spectrum, oracle_eval
mp.mp.dps = 50
cand = int(sys.argv[1]) if len(sys.argv) > 1 else 6761
w = workloads.config3(oracle_truth_spectrum, cpfit=False)
captured = []
orig = mo._PairChain.solve_lambda_system
def patched(self, cpfit=True, prec=1e-10, norm_eps=0.02):
    if self.mu[0] + self.mu[1] >= prec:
        captured.append(dict(mu=list(self.mu), lh=list(self.lh), T=self.T, P0=[list(map(float, self.P0[0])), list(map(float, self.P0[1]))]))
    return orig(self, cpfit, prec, norm_eps)
mo._PairChain.solve_lambda_system = patched
o = oracle_eval(w.times, w.lh, w.bands, w.pulses, w.flags, w.sample_date, float(w.split_time[cand]), list(w.params[cand]), w.jsfs[:1])
print('oracle status', o[2], 'migrating solves', len(captured))
st = captured[-1]
T = st['T']
mu = [st['mu'][0] * T, st['mu'][1] * T]
lh = [st['lh'][0] * T, st['lh'][1] * T]
P0 = st['P0']
ch = mo._PairChain()
ch.set_mu(*mu); ch.set_interval(lh, 1.0, P0)
x0 = np.array(lh)
def f_ref(x): return np.array(ch._residual_ect(list(x)))
def f_ex(x):
    x = [mp.mpf(float(v)) if not isinstance(v, mp.mpf) else v for v in x]
    M = mp.matrix([[-2*mp.mpf(mu[0]) - x[0], 0, mp.mpf(mu[1])], [0, -2*mp.mpf(mu[1]) - x[1], mp.mpf(mu[0])], [2*mp.mpf(mu[0]), 2*mp.mpf(mu[1]), -mp.mpf(mu[0]) - mp.mpf(mu[1])]])
    E = mp.expm(M); Mi = M**-1
    out = []
    for k in (0, 1):
        s = sum(mp.mpf(v) for v in P0[k]); pn = mp.matrix([mp.mpf(v)/s for v in P0[k]])
        vec1 = Mi*(Mi*((E - mp.eye(3))*pn)); vec2 = E*pn; pnc = sum(vec2); vec2 = Mi*vec2; vec = vec2 - vec1
        ect = (x[0]*vec[0] + x[1]*vec[1])/(1 - pnc)
        lam = mp.mpf(lh[k]); tgt = 1/lam - 1/(1/mp.e**(-lam) - 1)
        out.append(ect - tgt)
    return out
h = 1.4901161193847656e-08
def jac(f, x, conv=float):
    f0 = f(x); J = np.zeros((2, 2))
    for j in range(2):
        xx = np.array(x, dtype=float); xx[j] += h; dx = xx[j] - x[j]
        f1 = f(xx)
        for r in range(2): J[r, j] = conv((f1[r] - f0[r]) / dx)
    return np.array([conv(v) for v in f0]), J
fr, Jr = jac(f_ref, x0)
fe, Je = jac(f_ex, x0)
print('x0', x0, 'mu', mu)
print('f ref', fr, 'f exact', fe, 'rel diff', (fr - fe) / fe)
print('J ref\n', Jr, '\nJ exact-FD\n', Je, '\nrel', (Jr - Je) / Je)
pr = -np.linalg.solve(Jr, fr); pe = -np.linalg.solve(Je, fe)
print('GN step ref', pr, 'trial', x0 + pr); print('GN step exact', pe, 'trial', x0 + pe)
for name, xt in (('ref', x0 + pr), ('exact', x0 + pe)):
    fn = f_ref(xt); fx = f_ex(xt)
    print(name, 'trial cost ref-formula %.6e exact %.6e   (cost at x0 %.6e)' % (0.5*float(fn@fn), 0.5*float(sum(v*v for v in fx)), 0.5*float(fr@fr)))
# repeat the FD of the reference formula at a few nearby base points: is its Jacobian error systematic?
for d in (0.0, 1e-12, -1e-12, 3e-11, 1e-9):
    xx = x0 * (1 + d); _, J = jac(f_ref, xx); _, Jx = jac(f_ex, xx)
    print('base x0*(1%+.0e): J00 ref/exact - 1 = %+.3e   J01 %+.3e  J10 %+.3e  J11 %+.3e' % (d, J[0,0]/Jx[0,0]-1, J[0,1]/Jx[0,1]-1, J[1,0]/Jx[1,0]-1, J[1,1]/Jx[1,1]-1))
