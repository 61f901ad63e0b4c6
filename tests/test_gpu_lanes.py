"""misti_amd.lanes.LanePool: overlapped batches give exactly what one-at-a-time evaluation gives."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pool_matches_sequential_evaluation():
    from misti_amd import workloads
    from misti_amd.engine import Engine, truth_spectrum
    from misti_amd.lanes import LanePool
    w = workloads.config2(lambda *a: truth_spectrum(*a), n_split=16, n_rate=16)
    rng = np.random.default_rng(3)
    batches = []
    for k in range(24):                                   # different sub-grids and data rows per batch
        idx = rng.choice(w.n_cand, size=int(rng.integers(40, 200)), replace=False)
        rows = w.jsfs * (1.0 + 0.01 * k)
        batches.append((w.split_time[idx], w.params[idx], rows))
    with LanePool(w.times, w.lh, lanes=6, **w.engine_kwargs()) as pool:
        got = pool.map(batches)
        tk = pool.submit(*batches[0])
        assert tk.wait().done()
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        for (split, params, rows), (llk, jafs, status) in zip(batches, got):
            r = e.evaluate(split, params, rows)
            assert np.array_equal(r.status, status)
            assert np.array_equal(r.llk, llk, equal_nan=True) and np.array_equal(r.jafs, jafs, equal_nan=True)


def test_python_api_example_recovers_the_true_split():
    """examples/bootstrap_scan.py end to end in its own process: device-resident bootstrap scan + lane pool."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "bootstrap_scan.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "best split 64.00" in r.stdout and "40 scans on 8 lanes" in r.stdout
