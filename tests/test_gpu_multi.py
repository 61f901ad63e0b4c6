"""The multi-device form of the C ABI (misti_create_multi ...: SURVEY 8b's "variant taking a device list") on the one GPU of the
test box: the device list names device 0 two or three times, so the sharding - whole chains per context, one host thread each,
rows scattered into the caller's buffers - runs for real, and every output must equal the single-context call bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def grid():
    from misti_amd import workloads
    from misti_amd.engine import truth_spectrum
    w = workloads.config2(lambda *a: truth_spectrum(*a), n_split=12, n_rate=10, first_split=58, max_rate=0.5)
    w.jsfs = np.vstack([w.jsfs, w.jsfs * 0.5, w.jsfs * 0.25])
    return w


@pytest.mark.parametrize("devices", [(0,), (0, 0), (0, 0, 0)])
def test_multi_device_batch_equals_the_single_context(grid, devices):
    from misti_amd.engine import Engine, MultiEngine
    w = grid
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        one = e.evaluate(w.split_time, w.params, w.jsfs, want_lc=True, want_pr=True)
    with MultiEngine(w.times, w.lh, devices=devices, **w.engine_kwargs()) as m:
        got = m.evaluate(w.split_time, w.params, w.jsfs, want_lc=True, want_pr=True)
        cands, chains = m.last_shards()
    for k in ("llk", "jafs", "status", "lc"):
        assert np.array_equal(getattr(got, k), getattr(one, k), equal_nan=True), k
    # the pair-state trace: its last row carries work counters of the chain as the context saw it (batch-dependent), the rest is data
    assert np.array_equal(got.pr[:, :-1], one.pr[:, :-1], equal_nan=True)
    D = len(devices)
    assert sum(cands) == w.n_cand and sum(chains) == (10 if D > 1 else 0)          # ten rates = ten chains, none split between contexts
    if D > 1:
        assert max(chains) - min(chains) <= 1 and all(c == 12 * n for c, n in zip(cands, chains))


def test_multi_device_batch_without_parameters_and_with_bounds():
    """One chain for the whole batch (config-4 shape: candidates are interleaved over the contexts) and the README sweep with
    per-candidate band bounds (the chain key includes the bounds)."""
    from misti_amd import workloads
    from misti_amd.engine import Engine, MultiEngine, truth_spectrum
    w = workloads.config4(lambda *a: truth_spectrum(*a), n_split=21, n_rep=9)
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        one = e.evaluate(w.split_time, w.params, w.jsfs)
    with MultiEngine(w.times, w.lh, devices=(0, 0), **w.engine_kwargs()) as m:
        got = m.evaluate(w.split_time, w.params, w.jsfs)
        assert m.last_shards() == ([11, 10], [1, 1])
    assert np.array_equal(got.llk, one.llk, equal_nan=True) and np.array_equal(got.status, one.status)
    # per-candidate bounds
    w = workloads.config2(lambda *a: truth_spectrum(*a), n_split=4, n_rate=3, first_split=60, max_rate=0.3)
    bb = np.empty((w.n_cand, 1, 2), dtype=np.int32)
    bb[:, 0, 0] = 4 + (np.arange(w.n_cand) % 2) * 3          # two band starts: twice the chains
    bb[:, 0, 1] = -1
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        one = e.evaluate(w.split_time, w.params, w.jsfs, band_bounds=bb)
    with MultiEngine(w.times, w.lh, devices=(0, 0), **w.engine_kwargs()) as m:
        got = m.evaluate(w.split_time, w.params, w.jsfs, band_bounds=bb)
        assert sum(m.last_shards()[1]) == 6
    assert np.array_equal(got.llk, one.llk, equal_nan=True) and np.array_equal(got.jafs, one.jafs, equal_nan=True)


def test_multi_device_searches_equal_the_single_context():
    """misti_multi_nm_solve / misti_multi_basinhopping: contiguous blocks of starts per context, results start for start those of
    the single-context calls."""
    from misti_amd import workloads
    from misti_amd.engine import Engine, MultiEngine, truth_spectrum
    w = workloads.config3(lambda *a: truth_spectrum(*a), n_start=21)
    split = float(w.split_time[0])
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        nm1 = e.nm_solve(w.params, split, w.jsfs[0], tol=1e-4, maxiter=1000)
        bh1 = e.basinhopping(w.params[:7], split, w.jsfs[0], rngs=[40 + s for s in range(7)], niter=3, T=0.5, stepsize=0.05, interval=2)
    with MultiEngine(w.times, w.lh, devices=(0, 0, 0), **w.engine_kwargs()) as m:
        nm3 = m.nm_solve(w.params, split, w.jsfs[0], tol=1e-4, maxiter=1000)
        bh3 = m.basinhopping(w.params[:7], split, w.jsfs[0], rngs=[40 + s for s in range(7)], niter=3, T=0.5, stepsize=0.05, interval=2)
    for k in ("x", "llh", "nit", "nfev", "status"):
        assert np.array_equal(nm3[k], nm1[k], equal_nan=True), k
    for k in ("x", "llh", "nfev", "failures", "accepted"):
        assert np.array_equal(bh3[k], bh1[k], equal_nan=True), k


def test_multi_device_errors_are_reported_not_swallowed(grid):
    from misti_amd._lib import MistiError
    from misti_amd.engine import MultiEngine
    w = grid
    with pytest.raises(MistiError, match="device 99"):
        MultiEngine(w.times, w.lh, devices=(0, 99), **w.engine_kwargs())
    with pytest.raises(MistiError, match="empty"):
        MultiEngine(w.times, w.lh, devices=(), **w.engine_kwargs())


def test_c_example_multi_device(tmp_path):
    """examples/multi_device.c: the device-list form from plain C; two contexts on device 0 against one context, bit for bit, and
    SURVEY anchor A3's value for candidate 0."""
    import subprocess
    from test_host_cpu import _build_c_example
    exe = _build_c_example(tmp_path, "multi_device")
    r = subprocess.run([exe, "0", "0"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = dict(l.split(" = ") for l in r.stdout.splitlines() if " = " in l)
    assert int(out["contexts"]) == 2 and int(out["identical"]) == 1 and int(out["status"]) == 0
    assert abs(float(out["llh"]) - (-211.9189044185307)) <= 1e-9 * 212
    assert "6 candidates in 2 chains" in r.stdout


def test_multi_device_edge_shapes(grid):
    """Empty batch, spectrum-only (no replicates), and fewer chains than contexts (contexts without a candidate stay idle)."""
    from misti_amd.engine import Engine, MultiEngine
    w = grid
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e, MultiEngine(w.times, w.lh, devices=(0, 0, 0), **w.engine_kwargs()) as m:
        r = m.evaluate(np.zeros(0), np.zeros((0, 1)), w.jsfs)
        assert r.llk.shape == (0, 3) and r.status.shape == (0,) and m.last_shards() == ([0, 0, 0], [0, 0, 0])
        one = e.evaluate(w.split_time[:24], w.params[:24], None)
        got = m.evaluate(w.split_time[:24], w.params[:24], None)
        assert got.llk.shape == (24, 0) and np.array_equal(got.jafs, one.jafs, equal_nan=True) and np.array_equal(got.status, one.status)
        sel = np.arange(0, w.n_cand, 10)                               # one rate = one chain: 12 splits of it
        one = e.evaluate(w.split_time[sel], w.params[sel], w.jsfs)
        got = m.evaluate(w.split_time[sel], w.params[sel], w.jsfs)
        assert m.last_shards() == ([12, 0, 0], [1, 0, 0])
        assert np.array_equal(got.llk, one.llk, equal_nan=True)
