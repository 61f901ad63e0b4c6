#!/usr/bin/env python3
"""Clause 2b of the parity contract (tests/parity.py): the reference's own llh on inputs perturbed by 2^-44 (needs /root/reference: build
container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/wide_spread.py NAME [NAME ...]      ->  tests/golden/wide_spread.json (merged)

The contract's clause 2 measures the reference's indeterminacy under 2^-48 perturbations of its inputs.  On candidates whose llh has a
condition number of 1e8 ... 1e9 with respect to the inputs (default fit: solves that stop after one to three evaluations far from the root,
near-singular forward-difference Jacobians) the device's own rounding inside the path is worth MORE than a 2^-48 perturbation of the inputs -
a dozen ulps of an intermediate against 16 ulps of an input - and with the factor at 3 (round 5) such candidates fall out by factors 3 ... 10.
For exactly those candidates the reference is re-run here with perturbations of 2^-44 (256 ulps, 5.7e-14 relative: 17 000 times below the
1e-9 the north star asks of the llh): `spread_wide` = the largest relative change of ITS llh over kinds 0-15 of tests/parity.py: perturbed."""
import json
import os
import sys
import warnings

import numpy

numpy.mat = numpy.asmatrix
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import make_golden as mg               # noqa: E402  (imports the reference)
import parity                          # noqa: E402
from conftest import load_golden       # noqa: E402

LOG2 = -44
KINDS = 16


def main():
    names = sys.argv[1:]
    cases = {c["name"]: c for f in ("golden_small", "golden_synthetic", "golden_sweep", "golden_campaign", "golden_fullsize", "golden_default_fit",
                                    "golden_default_fit_256", "golden_fullsize_r05", "golden_config5_default_sample") if os.path.exists(os.path.join(HERE, f + ".json")) for c in load_golden(f)}
    path = os.path.join(HERE, "wide_spread.json")
    out = json.load(open(path))["cases"] if os.path.exists(path) else {}
    warnings.simplefilter("ignore")
    for n in names:
        i, o = cases[n]["in"], cases[n]["out"]
        parity.PERTURB = 2.0 ** LOG2
        try:
            vals = [mg.run_reference(*parity.perturbed(i["times"], i["lambdas"], k), i["sfs"], i["split"], i["mi"], i["pu"], i["kw"], i["params"])["llh"] for k in range(KINDS)]
        finally:
            parity.PERTURB = 2.0 ** -48
        fin = [v for v in vals if v is not None]
        spread = max(abs(v - o["llh"]) / abs(o["llh"]) for v in fin) if (fin and o["llh"] is not None) else None
        out[n] = {"perturb_log2": LOG2, "kinds": KINDS, "llh": vals, "spread_wide": spread, "fails": KINDS - len(fin)}
        print("%-24s spread at 2^-48 %.3g   at 2^%d %.3g   (%d of %d runs fail)" % (n, o.get("spread") or 0, LOG2, spread or 0, KINDS - len(fin), KINDS), flush=True)
    json.dump({"generator": "tests/golden/wide_spread.py", "scipy": "1.15.3", "numpy": "2.2.6", "cases": out}, open(path, "w"), indent=0)


if __name__ == "__main__":
    main()
