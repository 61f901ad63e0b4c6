#!/usr/bin/env python3
"""Where does the REFERENCE's default-fit solve of one two-population interval stall?  (build container only: imports /root/reference)

The stall rule of the HIP path (misti_kernels.hip: correct_body) returns the starting point where  rho = 0.29 W / (h max|J|) > 1,  W = eps / (1.4 min(d0, d1)^2) the
modelled noise width of the reference's residual, h = 1.5e-8 the forward-difference step, J the noise-free Jacobian at the start.  The reference's own traces in the
fixtures hold solves at rho >= 2 (205, all stalled) and rho <= 0.15 (15 000, 24 stalled) only.  Here the reference's CorrectLambda.SolveLambdaSystem(cpfit=False) is
run on ONE interval whose length T is swept (rates, migration and pair-state vectors drawn like BASELINE's grids), so that rho covers 0.02 ... 50, and each result is
compared with the root of the noise-free residual (mpmath, 40 digits).  Printed: per rho bin the fraction of solves that end closer to the START than to the ROOT.

    PYTHONDONTWRITEBYTECODE=1 python tools/stall_calibration.py > profiles/rNN_stall_calibration.txt"""
import contextlib
import io
import sys
import warnings

import numpy

numpy.mat = numpy.asmatrix
sys.path.insert(0, "/root/reference")
import mpmath as mp                      # noqa: E402
import CorrectLambda as CL               # noqa: E402

mp.mp.dps = 40
EPS = 2.220446049250313e-16
H = 1.4901161193847656e-08


def exact_residual(l, mu, P0, tgt):
    """The default-fit residual (CorrectLambda.py:94-110, 151-157) on the unit interval, in 40 digits."""
    M = mp.matrix([[-2 * mu[0] - l[0], 0, mu[1]], [0, -2 * mu[1] - l[1], mu[0]], [2 * mu[0], 2 * mu[1], -mu[0] - mu[1]]])
    E = mp.expm(M)
    Mi = M ** -1
    out = []
    for k in range(2):
        s = sum(P0[k])
        pn = mp.matrix([mp.mpf(v) / s for v in P0[k]])
        w = E * pn
        vec2 = Mi * w
        vec1 = Mi * (Mi * (w - pn))
        v = vec2 - vec1
        pnc = w[0] + w[1] + w[2]
        out.append((l[0] * v[0] + l[1] * v[1]) / (1 - pnc) - tgt[k])
    return out


def ect_one_pop(lam):
    return 1 / mp.mpf(lam) - 1 / (mp.e ** mp.mpf(lam) - 1)


def main():
    warnings.simplefilter("ignore")
    rng = numpy.random.default_rng(17)
    rows = []
    for draw in range(60):
        lh = [float(rng.uniform(0.6, 1.8)), float(rng.uniform(0.6, 1.8))]
        mu = [float(10 ** rng.uniform(-2.5, -0.5)), 0.0] if draw % 3 else [float(10 ** rng.uniform(-2.5, -0.5)), float(10 ** rng.uniform(-2.5, -0.5))]
        a, b = float(rng.uniform(0.2, 0.9)), float(rng.uniform(0.2, 0.9))
        P0 = [[a, 0.02 * rng.random(), 0.3 * rng.random()], [0.02 * rng.random(), b, 0.3 * rng.random()]]
        for T in 10 ** numpy.linspace(-4.6, -2.6, 25):
            T = float(T)
            c = CL.CorrectLambda()
            c.SetMu(mu[0], mu[1])
            c.SetInterval(list(lh), T, [list(P0[0]), list(P0[1])])
            with contextlib.redirect_stdout(io.StringIO()):
                try:
                    lc, _ = c.SolveLambdaSystem(cpfit=False)
                except SystemExit:
                    continue
            x_ref = [float(lc[0]) * T, float(lc[1]) * T]
            x0 = [lh[0] * T, lh[1] * T]
            mus = [mu[0] * T, mu[1] * T]
            tgt = [ect_one_pop(x0[0]), ect_one_pop(x0[1])]
            f0 = exact_residual(x0, mus, P0, tgt)
            J = [[0, 0], [0, 0]]
            for j in range(2):
                xs = list(x0)
                xs[j] += H
                fj = exact_residual(xs, mus, P0, tgt)
                for k in range(2):
                    J[k][j] = (fj[k] - f0[k]) / H
            # the noise-free root: Newton on the exact residual with the exact forward-difference Jacobian re-evaluated
            x = [mp.mpf(v) for v in x0]
            for it in range(12):
                f = exact_residual(x, mus, P0, tgt)
                Jx = mp.matrix(2, 2)
                for j in range(2):
                    xs = list(x)
                    xs[j] += mp.mpf(10) ** -12
                    fj = exact_residual(xs, mus, P0, tgt)
                    for k in range(2):
                        Jx[k, j] = (fj[k] - f[k]) / mp.mpf(10) ** -12
                dx = mp.lu_solve(Jx, mp.matrix([-f[0], -f[1]]))
                x = [x[0] + dx[0], x[1] + dx[1]]
                if max(abs(dx[0]), abs(dx[1])) < mp.mpf(10) ** -25:
                    break
            root = [float(x[0]), float(x[1])]
            dmin = min(2 * mus[0] + x0[0], 2 * mus[1] + x0[1])
            wm = (0.5 * EPS / 0.7) / (dmin * dmin)
            jmax = max(abs(float(J[k][j])) for k in range(2) for j in range(2))
            rho = 0.29 * wm / (H * jmax)
            k = 0 if abs(root[0] - x0[0]) / x0[0] >= abs(root[1] - x0[1]) / x0[1] else 1          # the coordinate that has to move
            need = (root[k] - x0[k]) / x0[k]
            got = (x_ref[k] - x0[k]) / x0[k]
            rows.append((rho, need, got, x_ref[0] <= 0 or x_ref[1] <= 0))
    rows.sort()
    print("# %d solves of /root/reference's CorrectLambda.SolveLambdaSystem(cpfit=False); rho = 0.29 W / (h max|J|) as the stall rule computes it" % len(rows))
    print("# stalled: the result is closer to the starting point than to the noise-free root (in the coordinate that has to move most; needed moves below 1e-6 skipped)")
    edges = [0, 0.02, 0.05, 0.1, 0.2, 0.35, 0.5, 0.7, 1.0, 1.4, 2.0, 3.0, 5.0, 10.0, 1e9]
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = [r for r in rows if lo <= r[0] < hi and abs(r[1]) > 1e-6]
        if not sel:
            continue
        stalled = sum(abs(r[2]) < 0.5 * abs(r[1]) for r in sel)
        neg = sum(r[3] for r in sel)
        print("rho in [%5.2f, %7.2f): %4d solves, %4d stalled (%3.0f %%), %3d ended at a non-positive rate; median needed move %.3g, median achieved/needed %.2f"
              % (lo, hi, len(sel), stalled, 100.0 * stalled / len(sel), neg, numpy.median([abs(r[1]) for r in sel]), numpy.median([r[2] / r[1] for r in sel])))


if __name__ == "__main__":
    main()
