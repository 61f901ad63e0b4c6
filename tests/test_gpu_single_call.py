"""One large sweep in ONE call: 16 config-2 grids with distinct rate axes - 65 536 candidates, 1 024
chains - run one chain per wave (speculation, trunk wave following), the workgroups pulling chains longest-first from a
queue.  Every output must be bit-identical to the 16 grids evaluated one call each, and to the packed launch shape."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _workloads():
    from misti_amd import workloads
    from misti_amd.engine import truth_spectrum
    spec = lambda *a: truth_spectrum(*a)
    return workloads.config2(spec), workloads.config2x16(spec)


def test_one_call_equals_sixteen_calls():
    from misti_amd.engine import Engine
    w1, w16 = _workloads()
    n1 = w1.n_cand
    assert w16.n_cand == 16 * n1 and len({tuple(p) for p in w16.params}) == 1024
    with Engine(w16.times, w16.lh, **w16.engine_kwargs()) as e:
        e.evaluate(w16.split_time, w16.params, w16.jsfs)                   # the launch shape follows the previous batch's chain count
        big = e.evaluate(w16.split_time, w16.params, w16.jsfs)
        work = e.evaluate(w16.split_time[:8], w16.params[:8], w16.jsfs, want_pr=True)
    assert (big.status == 0).mean() > 0.98
    with Engine(w1.times, w1.lh, **w1.engine_kwargs()) as e:
        for g in range(16):
            sl = slice(g * n1, (g + 1) * n1)
            r = e.evaluate(w16.split_time[sl], w16.params[sl], w16.jsfs)
            assert np.array_equal(r.status, big.status[sl]), g
            assert np.array_equal(r.llk, big.llk[sl], equal_nan=True), g
            assert np.array_equal(r.jafs, big.jafs[sl], equal_nan=True), g
    assert work.pr is not None


@pytest.mark.parametrize("env", [{"MISTI_FOLLOW_MAX_CHAINS": "1"}, {"MISTI_NO_FOLLOW": "1"}, {"MISTI_NO_TRUNK": "1"}],
                         ids=["packed", "trunk-after", "no-trunk"])
def test_queue_and_trimmed_trunk_equal_the_other_launch_shapes(env):
    """384 chains x 64 splits through the queue-fed one-chain-per-wave launch (trunk records stored only from the first interval a
    member reads) against the packed launch, the trunk built afterwards, and no trunk at all: same bits, rates and pair states included."""
    from misti_amd import workloads
    from misti_amd.engine import Engine, truth_spectrum
    w = workloads.config2x16(lambda *a: truth_spectrum(*a), n_grid=6)
    # a fractional split in every fourth candidate: tails and trimmed trunks together
    st = w.split_time.copy()
    st[::4] += 0.37
    out = {}
    for name, e_env in (("default", {}), ("other", env)):
        os.environ.update(e_env)
        try:
            with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
                e.evaluate(st, w.params, w.jsfs)
                out[name] = e.evaluate(st, w.params, w.jsfs, want_lc=True, want_pr=True)
        finally:
            for k in e_env:
                os.environ.pop(k, None)
    a, b = out["default"], out["other"]
    assert np.array_equal(a.status, b.status)
    for name in ("llk", "jafs", "lc"):
        assert np.array_equal(getattr(a, name), getattr(b, name), equal_nan=True), name
    assert np.array_equal(a.pr[:, :-1], b.pr[:, :-1], equal_nan=True)
    spec = a.pr[:, -1, 3]                                                   # work counters: solver steps taken from speculative slots
    assert spec.max() > 100                                                 # the default shape really was one chain per wave


def test_config5_shape_follows_with_more_chains_than_resident_workgroups():
    """2 048 chains (config 5: ancient sample, pulse) on 1 024 resident workgroups: two rounds through the queue (forced: beyond
    FOLLOW_MAX_CHAINS = 1 024 the default is the packed shape), every candidate against the packed shape."""
    from misti_amd import workloads
    from misti_amd.engine import Engine, truth_spectrum
    w = workloads.config5(lambda *a: truth_spectrum(*a))
    out = {}
    for name, env in (("follow", {"MISTI_FOLLOW_MAX_CHAINS": "4096"}), ("packed", {})):
        os.environ.update(env)
        try:
            with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
                e.evaluate(w.split_time, w.params, w.jsfs)
                out[name] = e.evaluate(w.split_time, w.params, w.jsfs)
        finally:
            for k in env:
                os.environ.pop(k, None)
    a, b = out["follow"], out["packed"]
    assert np.array_equal(a.status, b.status)
    assert np.array_equal(a.llk, b.llk, equal_nan=True)
    assert np.array_equal(a.jafs, b.jafs, equal_nan=True)


def test_busy_device_changes_the_launch_shape_not_the_results():
    """A 256-chain batch runs one chain per wave when the device is idle and packed ten per wave when three or more other
    contexts have batches in flight (misti_consts.h: FOLLOW_BUSY_*): 12 batches overlapped on 6 lanes against the same batch
    evaluated alone - every value bit for bit the same."""
    from misti_amd import workloads
    from misti_amd.engine import Engine, truth_spectrum
    from misti_amd.lanes import LanePool
    w = workloads.config2x16(lambda *a: truth_spectrum(*a), n_grid=4)
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        e.evaluate(w.split_time, w.params, w.jsfs)
        alone = e.evaluate(w.split_time, w.params, w.jsfs)
    with LanePool(w.times, w.lh, lanes=6, **w.engine_kwargs()) as pool:
        pool.map([(w.split_time, w.params, w.jsfs)] * 6)                      # every lane learns its chain count
        out = pool.map([(w.split_time, w.params, w.jsfs)] * 12)
    for llk, jafs, status in out:
        assert np.array_equal(status, alone.status)
        assert np.array_equal(llk, alone.llk, equal_nan=True) and np.array_equal(jafs, alone.jafs, equal_nan=True)


def test_busy_device_packs_small_default_fit_batches_without_changing_results():
    """The default fit's one-chain-per-wave kernel holds one wave per SIMD, so a 64-chain default-fit batch packs two chains per wave
    when three or more other contexts have batches in flight (run_dev, misti_api.cpp: 2.0 -> 2.6e7 evals/s on the headline grid with 20
    batches in flight) and keeps the latency shape alone: 12 batches overlapped on 6 lanes against the same batch alone, bit for bit."""
    from misti_amd import workloads
    from misti_amd.engine import Engine, truth_spectrum
    from misti_amd.lanes import LanePool
    w = workloads.config2(lambda *a: truth_spectrum(*a), n_split=16, n_rate=64, first_split=60, cpfit=False)
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        e.evaluate(w.split_time, w.params, w.jsfs)
        alone = e.evaluate(w.split_time, w.params, w.jsfs)
    with LanePool(w.times, w.lh, lanes=6, **w.engine_kwargs()) as pool:
        pool.map([(w.split_time, w.params, w.jsfs)] * 6)                      # every lane learns its chain count
        out = pool.map([(w.split_time, w.params, w.jsfs)] * 18)
    for llk, jafs, status in out:
        assert np.array_equal(status, alone.status)
        assert np.array_equal(llk, alone.llk, equal_nan=True) and np.array_equal(jafs, alone.jafs, equal_nan=True)
