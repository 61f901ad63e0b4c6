#!/usr/bin/env python3
"""Python API tour on synthetic data (needs an MI355X):

  1. a split-time scan x bootstrap replicates with the likelihood table kept on the device
     (`Engine.evaluate_dev` + `misti_argmax_dev` via `optimize.bootstrap_scan_dev`), i.e. what the bash loops of
     the reference's test.bs/*.sh + bs_conf_int.ipynb do with one MiSTI.py process per (split, replicate);
  2. many independent scans overlapped on a pool of lanes (`lanes.LanePool`).

    python examples/bootstrap_scan.py
"""
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from misti_amd import io as mio, synth                      # noqa: E402
from misti_amd.engine import Engine, truth_spectrum         # noqa: E402
from misti_amd.lanes import LanePool                        # noqa: E402
from misti_amd.optimize import bootstrap_scan_dev           # noqa: E402


def main():
    # two synthetic PSMC outputs -> merged grid (numT = 128), a truth with split index 64, data JSFS from its spectrum
    inp = mio.merge_psmc(mio.read_psmc_file(io.StringIO(synth.psmc_text(64, 1, synth.THETA_1))),
                         mio.read_psmc_file(io.StringIO(synth.psmc_text(65, 2, synth.THETA_2))))
    times, lh, _ = synth.self_consistent(inp, 64)
    jafs = truth_spectrum(times, lh, 64, [], [], 0)
    row = synth.counts_from_spectrum(jafs, 10 ** 6)
    table = np.array(mio.bootstrap_table(synth.chunk_rows(row, 20), 200))          # row 0 = the data, 200 resamples
    splits = np.arange(48, 81, dtype=float)

    with Engine(times, lh, cpfit=True, smooth=True) as e:
        t0 = time.perf_counter()
        mean, (lo, hi), best = bootstrap_scan_dev(e, splits, table)
        dt = time.perf_counter() - t0
    print("split scan %d values x %d replicates: best split %.2f, 95%% interval [%.2f, %.2f]  (%.1f ms, %d llk values)"
          % (len(splits), len(table), mean, lo, hi, 1e3 * dt, len(splits) * len(table)))

    # the same scan for 40 different resampled tables, overlapped on 8 lanes
    rng = np.random.default_rng(0)
    tables = [table[rng.integers(0, len(table), len(table))] for _ in range(40)]
    with LanePool(times, lh, lanes=8, cpfit=True, smooth=True) as pool:
        pool.map([(splits, None, tables[0])])                                      # contexts warm up
        t0 = time.perf_counter()
        out = pool.map([(splits, None, tb) for tb in tables])
        dt = time.perf_counter() - t0
    best = [splits[np.argmax(llk[:, 0])] for llk, _, _ in out]
    print("40 scans on 8 lanes: %.1f ms in total, best splits %s ..." % (1e3 * dt, best[:5]))


if __name__ == "__main__":
    main()
