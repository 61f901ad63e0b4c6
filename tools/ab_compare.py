#!/usr/bin/env python3
"""Bitwise A/B of two builds of the library (run on the GPU box).

    python -m misti_amd.build --out /tmp/variant.so -DSOMETHING        # a variant build
    MISTI_LIB_AB=1 MISTI_LIB=/tmp/variant.so python tools/ab_compare.py dump a.npz    # 300 random models + configs 2 and 3: llk, status, rates, spectra
    python tools/ab_compare.py dump b.npz                              # the in-tree build
    python tools/ab_compare.py cmp a.npz b.npz                         # arrays that differ in any bit

How the "bit-identical" claims of DESIGN.md section 4 were checked."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def dump(path):
    from random_campaign import random_batch
    from misti_amd import workloads
    from misti_amd.engine import Engine, truth_spectrum
    out = {}
    rng = np.random.default_rng(7)
    for i in range(int(os.environ.get("AB_MODELS", "300"))):
        c = random_batch(rng)
        with Engine(c["times"], c["lh"], c["bands"], c["pulses"], n_param=c["P"], sample_date=c["sd"], **c["flags"]) as e:
            r = e.evaluate(c["split"], c["params"], [c["sfs"]], want_lc=True)
        out["m%d_llk" % i] = r.llk; out["m%d_st" % i] = r.status; out["m%d_lc" % i] = r.lc; out["m%d_j" % i] = r.jafs
    for name in ("config2", "config3"):
        w = getattr(workloads, name)(lambda *a: truth_spectrum(*a))
        n = min(w.n_cand, 4096)
        with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
            r = e.evaluate(w.split_time[:n], w.params[:n], w.jsfs, want_lc=True)
        out[name + "_llk"] = r.llk; out[name + "_st"] = r.status; out[name + "_lc"] = r.lc; out[name + "_j"] = r.jafs
    np.savez(path, **out)
    print("saved", len(out), "arrays")


def cmp(pa, pb):
    a, b = np.load(pa), np.load(pb)
    bad = 0
    for k in a.files:
        x, y = a[k], b[k]
        if not (x.shape == y.shape and np.array_equal(x.view(np.uint8), y.view(np.uint8))):
            bad += 1
            if bad < 10:
                print("DIFF", k, np.nanmax(np.abs(x.astype(float) - y.astype(float))))
    print("arrays", len(a.files), "different", bad)
    return bad


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "dump":
        dump(sys.argv[2])
    elif len(sys.argv) >= 4 and sys.argv[1] == "cmp":
        sys.exit(1 if cmp(sys.argv[2], sys.argv[3]) else 0)
    else:
        print(__doc__)
