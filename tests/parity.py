"""Shared helpers of the parity tests: tolerance model and case plumbing."""
import math

import numpy as np

EPS = 2.220446049250313e-16
# Stated tolerance of the north star: log-likelihood within 1e-9 relative of the reference.
LLK_RTOL = 1e-9
# Rounding floor of the reference's own final formula  llh = llh_const + sum d_i log J_i
# (MigrationInference.py:600-609, :217-227): every summand is rounded to ~1 ulp of its own
# magnitude, and with 1e6 sites the summands are ~1e6..1e7 while llh can be ~ -20, so
# the reference itself is defined only up to a few ulps of the largest summand.
FLOOR_ULPS = 16
JAFS_RTOL = 1e-9
# corrected rates are an intermediate: where the residual of the correction is flat in one
# direction (pair all but coalesced) that component is undetermined at ~1e-6 although the
# likelihood is not (observed: <= 2.3e-9 everywhere else, 1.4e-6 in such a direction)
LC_RTOL = 1e-5


def llk_summand_scale(row, jafs, unfolded):
    d = [float(v) for v in row[1:]]
    n = sum(d)
    if unfolded:
        terms = [math.lgamma(n + 1)] + [math.lgamma(v + 1) for v in d] + [d[i] * abs(math.log(jafs[i])) for i in range(7)]
    else:
        f = [d[0] + d[6], d[1] + d[5], d[2] + d[4], d[3]]
        j = [jafs[0] + jafs[6], jafs[1] + jafs[5], jafs[2] + jafs[4], jafs[3]]
        terms = [math.lgamma(n + 1)] + [math.lgamma(v + 1) for v in f] + [f[i] * abs(math.log(j[i])) for i in range(4)]
    return sum(abs(t) for t in terms)


def llk_tol(ref_llk, row, jafs, unfolded):
    return LLK_RTOL * abs(ref_llk) + FLOOR_ULPS * EPS * llk_summand_scale(row, jafs, unfolded)


# Conditioning of the reference itself.  make_golden.py records `sens`: the factor by which
# a 2^-48 relative perturbation of the inputs is amplified in the reference's llh.  Normal
# candidates have sens ~1e3 (cancellation against llh_const); candidates whose lambda-correction
# ran into a flat residual (runaway corrected rate) have sens 1e6..1e10: there the reference's
# stopping point - and even whether it reports "correction failed" - is decided by rounding
# noise, so no independent implementation (nor the reference on another BLAS) reproduces it to
# 1e-9.  Parity at LLK_RTOL is asserted for determined candidates; for the others only
# agreement within the measured indeterminacy (or a failure status) is required.
PERTURB = 2.0 ** -48
SENS_DETERMINED = 3e4          # sens * 2^-48 < 1e-10


def determined(out):
    return out.get("sens") is not None and out["sens"] < SENS_DETERMINED


def loose_rtol(out):
    k = out.get("sens")
    # three perturbations sample the (discrete, flip-driven) indeterminacy only coarsely: allow 1000 x
    return 1e-2 if k is None else min(1e-2, max(LLK_RTOL, 1000.0 * k * PERTURB))


def engine_args(case_in):
    """Golden-case input -> keyword arguments of misti_amd.engine.MigrationInference."""
    i = case_in
    return (list(i["times"]), [list(x) for x in i["lambdas"]], list(i["sfs"]), i["split"],
            [list(x) for x in i["mi"]], [list(x) for x in i["pu"]]), dict(i["kw"])


def implemented(kw):
    """Modes the HIP path covers (grows as modes are added)."""
    return True
