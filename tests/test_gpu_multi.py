"""The multi-device form of the C ABI (misti_create_multi ...: SURVEY 8b's "variant taking a device list") on the one GPU of the
test box: the device list names device 0 two or three times, so the sharding - whole chains per context, one host thread each,
rows scattered into the caller's buffers - runs for real, and every output must equal the single-context call bit for bit."""
import os

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


# ---- round 5: cost-aware dealing, the gather inside the library, no exception across the ABI -------------------------------------
def test_chains_are_dealt_by_cost_and_results_do_not_change():
    """Chains of very different length (a grid ordered by split: every rate has its own largest split): the contexts' summed chain
    cost differs by less than one chain (longest-processing-time-first), the deal equals misti_amd.dist.chain_shards', and every
    output is still bit for bit the single context's."""
    from misti_amd import workloads
    from misti_amd.dist import chain_costs, chain_shards
    from misti_amd.engine import Engine, MultiEngine, truth_spectrum
    w = workloads.config2(lambda *a: truth_spectrum(*a), n_split=16, n_rate=9, first_split=40, max_rate=0.4)
    rate_of = np.unique(w.params[:, 0], return_inverse=True)[1]
    keep = w.split_time <= 43 + 1.5 * rate_of                                  # rate k keeps the splits up to 43 + 1.5 k
    keep &= ~((rate_of == 3) & (w.split_time > 41))
    split, params = w.split_time[keep].copy(), w.params[keep]
    split[(rate_of[keep] == 5) & (split == split[rate_of[keep] == 5].max())] -= 0.5      # one chain ends on a fractional split
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        one = e.evaluate(split, params, w.jsfs, want_lc=True)
    for D in (2, 3):
        with MultiEngine(w.times, w.lh, devices=(0,) * D, **w.engine_kwargs()) as m:
            got = m.evaluate(split, params, w.jsfs, want_lc=True)
            cands, chains = m.last_shards()
            cost = m.last_cost()
        for k in ("llk", "jafs", "status", "lc"):
            assert np.array_equal(getattr(got, k), getattr(one, k), equal_nan=True), (D, k)
        chain, c = chain_costs(params, len(split), split)
        assert sum(chains) == 9 and max(cost) - min(cost) <= c.max() and abs(sum(cost) - c.sum()) < 1e-9
        sh = chain_shards(params, len(split), D, split)
        assert [len(s) for s in sh] == cands
        assert np.allclose(sorted(cost), sorted(c[np.unique(chain[s])].sum() for s in sh))


def test_gather_inside_the_library_on_a_one_device_communicator():
    """misti_multi_eval_batch_dev on the device list {0}: device pointers in, the log-likelihoods and statuses all-gathered by RCCL
    (ncclAllGather on a communicator made by ncclCommInitAll) inside the library - the table a C caller gets without PCIe.  On one
    device the gather is the identity, but the communicator, the grouped collective on the context's stream and the padding are real."""
    import torch
    from misti_amd import workloads
    from misti_amd._lib import MistiError
    from misti_amd.engine import Engine, MultiEngine, truth_spectrum
    w = workloads.config2(lambda *a: truth_spectrum(*a), n_split=6, n_rate=5, first_split=60, max_rate=0.5)
    w.jsfs = np.vstack([w.jsfs, w.jsfs * 0.5])
    n, R, per = w.n_cand, 2, w.n_cand + 3
    dev = torch.device("cuda", 0)
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        one = e.evaluate(w.split_time, w.params, w.jsfs)
    d_split = torch.as_tensor(w.split_time, device=dev)
    d_par = torch.as_tensor(w.params, device=dev)
    d_jsfs = torch.as_tensor(w.jsfs, device=dev)
    d_all = torch.zeros((1, per, R), dtype=torch.float64, device=dev)
    d_st = torch.zeros((1, per), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with MultiEngine(w.times, w.lh, devices=(0,), **w.engine_kwargs()) as m:
        for _ in range(2):                                                      # the communicator is made once and reused
            m.evaluate_dev_gathered([n], per, [d_split.data_ptr()], [d_par.data_ptr()], R, [d_jsfs.data_ptr()], [d_all.data_ptr()], [d_st.data_ptr()])
            m.sync()
        got, st = d_all.cpu().numpy(), d_st.cpu().numpy()
        assert np.array_equal(got[0, :n], one.llk, equal_nan=True) and np.isnan(got[0, n:]).all()
        assert np.array_equal(st[0, :n], one.status) and (st[0, n:] == -1).all()
        with pytest.raises(MistiError, match="do not fit"):
            m.evaluate_dev_gathered([per + 1], per, [d_split.data_ptr()], [d_par.data_ptr()], R, [d_jsfs.data_ptr()], [d_all.data_ptr()])
    with MultiEngine(w.times, w.lh, devices=(0, 0), **w.engine_kwargs()) as m2:
        with pytest.raises(MistiError, match="listed twice"):
            m2.evaluate_dev_gathered([n, 0], per, [d_split.data_ptr()] * 2, [d_par.data_ptr()] * 2, R, [d_jsfs.data_ptr()] * 2, [d_all.data_ptr()] * 2)


def test_gathered_form_on_three_contexts_through_the_rccl_double(tmp_path):
    """misti_multi_eval_batch_dev with D > 1 before a multi-GPU node ever runs it.  tests/multi_host/fake_rccl.cpp (built
    here with hipcc; it copies between the ranks' tables on their streams, ordered by events) is bound through MISTI_RCCL_LIB in a child
    process, and the device list {0, 0, 0} runs three contexts, three persistent workers and ONE grouped ncclAllGather over three
    communicators: ragged and empty shards, NaN / -1 padding, every context's table equal to one context's misti_eval_batch bit for bit,
    a worker that throws.  Real RCCL over xGMI with D > 1 stays unmeasured on hardware."""
    import shutil
    import subprocess
    import sys
    from conftest import ROOT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    double = str(tmp_path / "libfake_rccl_hip.so")
    subprocess.run([hipcc, "-O1", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950",
                    os.path.join(ROOT, "tests", "multi_host", "fake_rccl.cpp"), "-o", double], check=True, timeout=300)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multi_host", "gather_double_check.py")], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, MISTI_RCCL_LIB=double))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = dict(l.split(" = ") for l in r.stdout.splitlines() if " = " in l)
    assert out["identical"] == "1" and out["throw"] == "reported", r.stdout
    assert int(out["rounds"]) == 5 and int(out["collectives"]) == 2 * 2 * 5 and int(out["finite"]) > 100, r.stdout   # llk + status, twice per layout


def test_an_exception_in_a_worker_thread_fails_the_call_not_the_process(grid):
    """include/misti_hip.h: no C++ exception crosses the ABI.  A worker thread that throws (internal test hook
    misti_multi_test_throw_in_worker_: not in the public header, nothing in the library reads the environment for it) makes the call
    return MISTI_E_ARG with the context's message; the persistent workers survive, and with the hook cleared the SAME object works."""
    import ctypes as C
    from misti_amd._lib import MistiError
    from misti_amd.engine import MultiEngine
    w = grid
    with MultiEngine(w.times, w.lh, devices=(0, 0), **w.engine_kwargs()) as m:
        want = m.evaluate(w.split_time, w.params, w.jsfs)
        hook = m._lib.misti_multi_test_throw_in_worker_
        hook.restype, hook.argtypes = C.c_int, [C.c_void_p, C.c_int]
        assert hook(m._m, 1) == 0
        for _ in range(2):
            with pytest.raises(MistiError, match=r"context 1 of 2.*misti_multi_test_throw_in_worker_") as ei:
                m.evaluate(w.split_time, w.params, w.jsfs)
            assert ei.value.code == -1
        assert hook(m._m, -1) == 0
        got = m.evaluate(w.split_time, w.params, w.jsfs)
        assert np.array_equal(got.llk, want.llk, equal_nan=True) and np.array_equal(got.status, want.status)


def test_c_example_gathered_on_the_devices(tmp_path):
    """examples/multi_device_gather.c: misti_multi_eval_batch_dev + misti_multi_sync from plain C with its own device buffers (HIP runtime C
    entry points) - the table gathered by RCCL inside the library equals one context's misti_eval_batch bit for bit, padding rows are NaN / -1."""
    import subprocess
    from test_host_cpu import _build_c_example
    exe = _build_c_example(tmp_path, "multi_device_gather", hip_runtime=True)
    r = subprocess.run([exe, "0"], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr
    out = dict(l.split(" = ") for l in r.stdout.splitlines() if " = " in l)
    assert int(out["contexts"]) == 1 and int(out["identical"]) == 1 and int(out["rows_per_shard"]) == 12
    assert abs(float(out["llh"]) - (-211.9189044185307)) <= 1e-9 * 212
