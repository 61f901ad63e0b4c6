#!/usr/bin/env python3
"""One candidate of a BASELINE workload through the HIP path with the solver trace on, and through the oracle with every
least_squares call logged: the first interval whose (nfev, status) differ, with both iteration histories side by side
(run on the GPU box).

    python tools/trace_candidate.py CANDIDATE [WORKLOAD]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
warnings.simplefilter('ignore')
import numpy as np
from scipy import optimize
from misti_amd import workloads
from misti_amd.engine import Engine, truth_spectrum
import oracle.misti_oracle as mo
from oracle.batch import oracle_eval
wl = sys.argv[2] if len(sys.argv) > 2 else 'config2'       # config2 | config3 | config5 | config3:default ...
name, _, fit = wl.partition(':')
w = getattr(workloads, name)(lambda *a: truth_spectrum(*a), **({'cpfit': fit == 'cpfit'} if fit else {}))
c = int(sys.argv[1]) if len(sys.argv) > 1 else 3427
s, p = float(w.split_time[c]), list(w.params[c])
# oracle with logged least_squares calls
calls = []
orig = optimize.least_squares
def logged(fun, x0, *a, **kw):
    pts = []
    def f(x, *aa, **kk):
        r = fun(x, *aa, **kk); pts.append((tuple(np.atleast_1d(x)), tuple(np.atleast_1d(r)))); return r
    res = orig(f, x0, *a, **kw)
    lb = np.atleast_1d(kw.get('bounds', (-np.inf, np.inf))[0])
    calls.append(dict(x0=tuple(np.atleast_1d(x0)), nfev=res.nfev, status=res.status, x=tuple(res.x), pts=pts, bounded=bool(np.isfinite(lb).any())))
    return res
mo.optimize.least_squares = logged
try:
    o = oracle_eval(w.times, w.lh, w.bands, w.pulses, w.flags, w.sample_date, s, p, w.jsfs)
finally:
    mo.optimize.least_squares = orig
print('oracle llk', o[0][0] if o[2] == 0 else None, 'solves', len(calls))
with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
    e.enable_solver_trace(True)
    r = e.evaluate([s], [p], w.jsfs, want_lc=True)
    tr = e.solver_trace(1, cand=0)
print('hip llk', r.llk[0, 0], 'status', int(r.status[0]), 'oracle status', o[2], 'rel', abs(r.llk[0,0]-o[0][0])/abs(o[0][0]) if o[2] == 0 else None)
k3 = np.where(tr["kind"][0] == 3)[0]
print('hip unbounded solves', len(k3))
# oracle's two-pop migrating solves are the calls with 2 unknowns and infinite bounds: in order of intervals
mig = [cl for cl in calls if len(cl['x0']) == 2 and not cl['bounded']]      # the default fit's no-migration intervals are bounded 2-unknown solves
print('oracle 2-unknown solves', len(mig))
ALL = '--all' in sys.argv          # do not stop at the first interval that differs: one line per differing interval, iterates for the oracle's last solve
for j, t in enumerate(k3):
    if j >= len(mig):
        print('interval', t, ': the oracle stopped before it (failed at its solve %d); hip lc' % (len(mig) - 1), r.lc[0, max(0, t - 2):t + 1].tolist())
        break
    cl = mig[j]
    hn, hs = int(tr["nfev"][0, t]), int(tr["status"][0, t])
    flag = '' if (hn, hs) == (cl['nfev'], cl['status']) else '   <<<<< DIFFERS'
    last = j == len(mig) - 1
    if flag or cl['nfev'] > 8 or last:
        print('interval', t, 'oracle nfev/status', cl['nfev'], cl['status'], 'hip', hn, hs, 'x oracle', cl['x'], 'hip lc x T', (r.lc[0, t] * w.times[t]).tolist(), flag)
    if (flag and not ALL) or (ALL and last):
        it = tr["iterates"][t]
        # oracle trial points: every third evaluation (base, +h0, +h1) -> unique base points
        base = []
        for x, f in cl['pts']:
            if not base or (abs(x[0] - base[-1][0][0]) > 3e-8 * max(1, abs(x[0])) or abs(x[1] - base[-1][0][1]) > 3e-8 * max(1, abs(x[1]))):
                base.append((x, f))
        for i in range(max(len(base), hn)):
            ox = base[i][0] if i < len(base) else None
            hx = tuple(it[i]) if i < it.shape[0] and np.isfinite(it[i]).all() else None
            rel = max(abs(ox[k] - hx[k]) / max(abs(ox[k]), 1e-300) for k in (0, 1)) if ox and hx else None
            print('   it', i, 'oracle', ox, 'f', base[i][1] if i < len(base) else None, 'hip', hx, 'rel', rel)
        if not ALL:
            break
