#!/usr/bin/env python3
"""Table of HIP-vs-reference discrepancies over the golden vectors (run on the GPU box).

    python tools/parity_report.py [golden_small golden_fullsize ...] > profiles/rNN_parity_goldens.txt

Per case: relative llk error, the clauses of the contract (tests/parity.py) - the 1e-9 tolerance (+ rounding floor)
and SELF_FACTOR (3) x the reference's own measured indeterminacy (`spread`: largest relative change of the reference's llh under
2^-48 input perturbations, 3 kinds for determined cases, 9 for the others, 64 for the campaign's and the full-size outliers', 16 for the
default-fit cases at numT = 128; `internal`: the same under one ulp in its own pair-chain matrix exponential, 16 runs) - which clause the case falls under and the FACTOR
err / spread (or err / internal) for it; max relative JAFS and lc errors."""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import load_golden                                   # noqa: E402
from parity import SELF_FACTOR, branch_of, branch_statistics, chain_key, fixed_in_advance_names, engine_args, internal_of, llk_tol, minority_tail, spread_of, status_flips_wide, wide_of    # noqa: E402
from misti_amd.engine import MigrationInference                    # noqa: E402


def main():
    rows = []
    n_tight = n_self = n_int = n_wide = n_out = n_fail_ok = n_fail_bad = 0
    worst_factor = worst_int = worst_wide = 0.0
    factors = []
    classified = {}        # case -> (chain key, branch record)
    files = [a for a in sys.argv[1:] if not a.startswith("-")] or ["golden_small", "golden_synthetic", "golden_sweep", "golden_campaign", "golden_fullsize", "golden_default_fit",
                                                                    "golden_default_fit_256", "golden_fullsize_r05", "golden_config5_default_sample", "golden_config2b", "golden_config2b_allchains", "golden_config3b_fixed64", "golden_config5b_fixed64", "golden_config5b_default_fixed64", "golden_config3b_default_fixed64", "golden_config2c", "golden_config3b", "golden_config2n255", "golden_config2u", "golden_config2m", "golden_config2f"]
    for f in files:
        for c in load_golden(f, optional=True):
            o = c["out"]
            args, kw = engine_args(c["in"])
            with contextlib.redirect_stdout(io.StringIO()):
                m = MigrationInference(*args, **kw)
                llh = m.JAFSLikelihood(list(c["in"]["params"]))
            if o["llh"] is None or llh == -np.inf:
                both = o["llh"] is None and llh == -np.inf
                flips = (o.get("pert_finite", 0) > 0 or o.get("internal_finite", 0) > 0) if o["llh"] is None else (o.get("pert_fail", 0) > 0 or o.get("internal_fail", 0) > 0)
                wide_flip = o["llh"] is None and status_flips_wide(c["name"])
                ok = both or flips or wide_flip
                n_fail_ok += ok
                n_fail_bad += not ok
                rows.append((c["name"], "ref -inf" if o["llh"] is None else "%.6g" % o["llh"], "hip -inf" if llh == -np.inf else "%.6g" % llh,
                             "", "", "", "", "", "", "", "both fail" if both else ("reference flips" if flips else ("reference flips at 2^-40 ... 2^-32" if wide_flip else "MISMATCH"))))
                continue
            err = abs(llh - o["llh"]) / abs(o["llh"])
            tol = llk_tol(o["llh"], c["in"]["sfs"], o["JAFS"], bool(kw.get("unfolded"))) / abs(o["llh"])
            spread = spread_of(o)
            internal = internal_of(o)
            wide = wide_of(o)
            ej = np.max(np.abs(np.array(m.JAFS) / np.array(o["JAFS"]) - 1))
            el = np.max(np.abs(np.array(m.lc) / np.array(o["lc"]) - 1))
            if err <= tol:
                cls, factor = "1e-9", ""
                n_tight += 1
            elif spread is not None and err <= SELF_FACTOR * spread:
                cls, factor = "self", "%.2f" % (err / spread)
                worst_factor = max(worst_factor, err / spread)
                factors.append(err / spread)
                n_self += 1
            elif internal is not None and err <= SELF_FACTOR * internal:
                cls, factor = "internal", "%.2f" % (err / internal)
                worst_int = max(worst_int, err / internal)
                factors.append(err / internal)
                n_int += 1
            elif wide is not None and err <= SELF_FACTOR * wide:
                cls, factor = "wide (2b)", "%.2f of its 2^-44 spread %.2e; %.1f x its 2^-48 spread" % (err / wide, wide, err / max(spread or 0.0, internal or 0.0, 1e-300))
                worst_wide = max(worst_wide, err / wide)
                n_wide += 1
            else:
                cls, factor = "OUTSIDE", "%.2f" % (err / max(spread or 0.0, internal or 0.0)) if (spread or internal) else "inf"
                n_out += 1
            # which branch of the reference's own runs the device is on (tests/parity.py: branch_of): "maj 61/64" = the branch 61 of its 64 finite
            # runs are on; "min 3/64" = a minority branch; "none" = on no branch of the reference; "1 mode" = the reference's runs are not bimodal
            b = branch_of(o, llh)
            if b is not None:
                classified[c["name"]] = (chain_key(c), b)
            if b is None:
                br = "-"
            elif b["n_modes"] < 2:
                br = "1 mode" if b["mode"] == 0 else "1 mode, off it"
            else:
                br = "none/%d modes" % b["n_modes"] if b["mode"] is None else "%s %d/%d" % ("maj" if b["mode"] == 0 else "min", round(b["share"] * b["runs"]), b["runs"])
                pass
            rows.append((c["name"], "%.2e" % err, "%.2e" % tol, "%.2e" % spread if spread is not None else "-", "%.2e" % internal if internal is not None else "-", factor,
                         "%.1e" % ej, "%.1e" % el, "%d/%d" % (o.get("pert_fail", 0), len(o.get("pert_llh", []))), br, cls))
    w = max(len(r[0]) for r in rows)
    print("%-*s %10s %10s %10s %10s %7s %9s %9s %6s %-14s %s" % (w, "case", "llk rel", "tol 1e-9", "ref spread", "internal", "factor", "JAFS rel", "lc rel", "pfail", "branch", "clause"))
    for r in rows:
        print("%-*s %10s %10s %10s %10s %7s %9s %9s %6s %-14s %s" % ((w,) + r))
    print()
    print("finite on both sides: %d within 1e-9 (+ floor), %d within %g x the reference's own spread under input perturbations (worst factor %.2f),"
          % (n_tight, n_self, SELF_FACTOR, worst_factor))
    print("    %d more within %g x its spread under one ulp in its own expm (worst factor %.2f)," % (n_int, SELF_FACTOR, worst_int))
    print("    %d under clause 2b only (within %g x its spread under 2^-44 input perturbations, worst factor %.2f; each listed above with its factor against the 2^-48 spreads),"
          % (n_wide, SELF_FACTOR, worst_wide))
    print("    %d OUTSIDE the contract" % n_out)
    if factors:
        f = np.sort(np.array(factors))
        print("factor used under clause 2 (%d candidates): median %.2f, 90 %% %.2f, 99 %% %.2f, max %.2f; %d within 1 x, %d within 3 x"
              % (len(f), np.quantile(f, 0.5), np.quantile(f, 0.9), np.quantile(f, 0.99), f.max(), int((f <= 1).sum()), int((f <= 3).sum())))
    print("failures: %d agree or are reference flips, %d mismatches" % (n_fail_ok, n_fail_bad))
    if classified:
        # the mode-aware reading of clause 2, per CHAIN: members of a chain inherit the flip of one of its solves
        fixed_names = fixed_in_advance_names()
        for title, names in (("FIXED IN ADVANCE (every chain of the held-out grid, the evenly spaced default-fit candidates, the README sweep, the small fixtures)", fixed_names),
                             ("SELECTED because the device deviated from the checker (reported, never asserted: being off the majority is what selected them)", set(classified) - fixed_names)):
            st = branch_statistics(classified, names)
            print()
            print("branches, cases %s:" % title)
            print("    %d chains whose reference runs are bimodal; the device is on the reference's MAJORITY branch on %d of them, off it on %d;"
                  % (st["bimodal_chains"], st["bimodal_chains"] - st["on_minority"], st["on_minority"]))
            print("    the reference's own runs are off their majority with summed frequency %.1f over those chains: P(at least %d by chance) = %.3g, P(at most %d) = %.3g"
                  % (st["expected"], st["on_minority"], st["tail"], st["on_minority"], st["tail_low"]))
            for name, n, p in st["detail"]:
                print("    chain of %-28s %2d member(s) off the majority; the reference's own runs are off it with frequency %.3f there" % (name, n, p))

if __name__ == "__main__":
    main()
