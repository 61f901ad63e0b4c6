#!/usr/bin/env python3
"""Second pass of the full-size parity check (CPU only): every candidate the first pass (tools/fullsize_report.py --dump: the HIP path's value of
EVERY candidate against the compiled baseline, 16 + 16 perturbed runs, SELF_FACTOR) put outside the contract or flagged as a status mismatch,
against /root/reference ITSELF - its value and its own spreads from the reference-run goldens (tests/golden/golden_fullsize*.json,
golden_default_fit*.json) - under the contract of tests/parity.py.

    python tools/fullsize_second_pass.py gpurun_out/fs5 > profiles/rNN_fullsize_contract.txt"""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden                                                      # noqa: E402
from parity import SELF_FACTOR, internal_of, llk_bound, spread_of, status_flips_wide, wide_of   # noqa: E402


def main():
    dump = sys.argv[1]
    ref = {}
    for f in ("golden_fullsize", "golden_default_fit", "golden_default_fit_256", "golden_fullsize_r05", "golden_config2b", "golden_config2c", "golden_config3b", "golden_config2n255", "golden_config2u", "golden_config2m", "golden_config2f"):
        for c in load_golden(f):
            ref[(c["fullsize"]["workload"], int(c["fullsize"]["cand"]))] = c
    print("# first pass: tools/fullsize_report.py (every candidate, compiled baseline as checker, 16 + 16 runs, factor %g) - %s/report.txt" % (SELF_FACTOR, dump))
    print("# second pass: each flagged candidate against /root/reference itself (value, 2^-48 input spread, one-ulp-in-expm spread; clause 2b where measured)")
    for f in sorted(glob.glob(os.path.join(dump, "fullsize_*.json"))):
        j = json.load(open(f))
        wl = j["workload"]
        z = np.load(f.replace(".json", ".npz"))
        pos = {int(c): k for k, c in enumerate(z["idx"])}
        flagged = [o[0] for o in j["outside"]] + [m[0] for m in j["mismatch"]]
        n_by = {}
        rows, unstudied, outside = [], [], []
        for cand in flagged:
            c = ref.get((wl, cand))
            if c is None:
                unstudied.append(cand)
                continue
            o = c["out"]
            h, hs = float(z["hip_llk"][pos[cand]]), int(z["hip_status"][pos[cand]])
            if o["llh"] is None or hs != 0:
                if (o["llh"] is None) == (hs != 0):
                    cls = "both fail"
                else:
                    flips = (o.get("pert_finite", 0) > 0 or o.get("internal_finite", 0) > 0) if o["llh"] is None else (o.get("pert_fail", 0) > 0 or o.get("internal_fail", 0) > 0)
                    cls = "reference flips (2^-48 / one ulp)" if flips else ("reference flips (2^-40 ... 2^-32)" if status_flips_wide(c["name"]) else "STATUS MISMATCH")
                    if cls == "STATUS MISMATCH":
                        outside.append(cand)
                n_by[cls] = n_by.get(cls, 0) + 1
                rows.append((cand, cls, "", ""))
                continue
            bound, clause = llk_bound(o["llh"], c["in"]["sfs"], o["JAFS"], bool(c["in"]["kw"].get("unfolded")), spread_of(o), internal_of(o), wide_of(o))
            err = abs(h - o["llh"])
            sp = max(spread_of(o) or 0.0, internal_of(o) or 0.0)
            if err <= 1e-9 * abs(o["llh"]):
                cls = "1e-9"
            elif err <= bound:
                cls = clause
            else:
                cls = "OUTSIDE"
                outside.append(cand)
            n_by[cls] = n_by.get(cls, 0) + 1
            rows.append((cand, cls, "%.3g" % (err / abs(o["llh"])), "%.2f" % (err / abs(o["llh"]) / sp) if sp else "-"))
        print()
        print("%s: %d candidates, %d with a value on both sides in the first pass, %d within 1e-9 of the baseline, %d within %g x their baseline spread;"
              % (wl, j["n"], j["both"], j["tight"], j["self"], SELF_FACTOR))
        print("   first pass flags %d (%d outside, %d status mismatches); second pass against the REFERENCE: %s; unstudied %s; OUTSIDE %s"
              % (len(flagged), len(j["outside"]), len(j["mismatch"]), ", ".join("%d %s" % (v, k) for k, v in sorted(n_by.items())) or "-", unstudied or "none", outside or "none"))
        for r in rows:
            print("      candidate %6d  %-34s rel %-9s factor of the reference's own spread %s" % r)


if __name__ == "__main__":
    main()
