#!/usr/bin/env python3
"""Expected JSFS and likelihood of a fully specified ``ms`` model - the reference's
``TestModel.py`` (``/root/reference/TestModel.py:36-120``) on the HIP engine.

    python -m misti_amd.testmodel "4 100 -t ... -I 2 2 2 -n 1 10 -em 0.0 1 2 2.0 -ej 0.045 2 1 ..." [-j data.sfs] [-uf] [-o out]

The ms command line is read by ``misti_amd.io.read_ms``; its rates are the TRUE ones
(``trueEPS``), so no lambda-correction runs.  Printed: the expected spectrum, with ``-j`` also
the data spectrum, the data's likelihood under the model and the maximum of the likelihood
function.  The forward map (``CoalescentRates``) then gives the rates PSMC would infer, and
``-o`` writes model + forward rates in the ``#MiSTI2`` format.  ``--bssize`` of the reference
is not offered (its loop refers to an undefined name, TestModel.py:112).
"""
from __future__ import annotations

import argparse
import sys

from . import io as mio
from .engine import MigrationInference


def build_parser():
    p = argparse.ArgumentParser(description="Expected JSFS / likelihood of an ms model (MI355X engine).")
    p.add_argument("msstring", help="ms command line (one string)")
    p.add_argument("-j", "--fjafs", default="", help="joint allele frequency spectrum file")
    p.add_argument("-o", "--fout", default="", help="output file for the model with forward rates")
    p.add_argument("-uf", action="store_true", help="unfolded spectrum")
    p.add_argument("--funits", default="setunits.txt", help="file with units")
    p.add_argument("--device", type=int, default=0, help="HIP device index")
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    units = mio.Units.from_file(a.funits)
    print(units.describe())
    have_data = a.fjafs != ""
    if have_data:
        rows, _, _ = mio.read_jsfs(a.fjafs)
        sfs = [sum(r[i] for r in rows) for i in range(8)]
    else:
        sfs = [1 for _ in range(8)]
    print("WARNING: read_ms() trusts the ms command line (two populations, one -ej).", file=sys.stderr)
    d = mio.read_ms(a.msstring)
    mig = MigrationInference(d.times, d.lambdas, sfs, d.divergenceTime, d.mi, d.pu, unfolded=a.uf, trueEPS=True, device=a.device)
    llh = mig.JAFSLikelihood([])
    print("Expected SFS", mig.JAFS)
    if have_data:
        norm = sum(sfs[1:])
        print("Data     SFS", [v / norm for v in sfs[1:]])
        print("data llh under the model is", llh)
        print("maximum of the llh function is", mig.MaximumLLHFunction())
    mig.CoalescentRates()
    if a.fout:
        with open(a.fout, "w") as f:
            f.write(mio.format_migration(mig, llh, 2 * units.N0, 1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
