"""Multi-rank path on CPU: world_size 2 (and 3) gloo processes shard ONE candidate grid with
misti_amd.dist.evaluate_sharded and gather the log-likelihoods.  The per-rank evaluator is engine-shaped: the
oracle on a tiny grid, returning a BatchResult-like object, with the engine's chain-sharing semantics (a rank's
candidates with identical parameters share one lambda-correction up to the largest split among THAT RANK's
members) - so what is checked is that sharding a grid whose chains span ranks changes no value."""
import os
import socket
import sys
import warnings

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def tiny_grid():
    """numT = 10; 4 split values (one fractional) x 3 rates of one band following the split; 2 replicates."""
    times = [0.02, 0.03, 0.05, 0.05, 0.08, 0.1, 0.15, 0.2, 0.3]
    lh = [[1.0, 1.6], [1.0, 1.6], [0.8, 1.2], [0.8, 1.2], [1.1, 0.9], [1.1, 0.9], [0.9, 0.9], [0.9, 0.7], [0.8, 0.7], [0.7, 0.7]]
    bands = [(0, 1, -1, 0.0, 0)]
    flags = dict(cpfit=True, true_eps=False, smooth=True, unfolded=False)
    split = np.repeat(np.array([4.0, 5.0, 5.5, 7.0]), 3)
    params = np.tile(np.array([[0.05], [0.3], [1.2]]), (4, 1))
    jsfs = np.array([[100000, 900, 250, 1000, 600, 400, 260, 410], [100000, 880, 262, 1011, 590, 395, 270, 400]], dtype=float)
    return times, lh, bands, flags, split, params, jsfs


class OracleEngine:
    """Engine-shaped evaluator on the CPU (the oracle): `evaluate(split, params, jsfs) -> object with .llk, .status`.
    Like the HIP engine it evaluates the batch it is GIVEN: chains are formed among these candidates only."""

    def __init__(self):
        self.times, self.lh, self.bands, self.flags = tiny_grid()[:4]
        self.chains_seen = None

    def evaluate(self, split, params, jsfs):
        from types import SimpleNamespace
        from oracle.batch import oracle_eval
        keys = {tuple(p) for p in params}
        self.chains_seen = len(keys)
        llk = np.empty((len(split), len(jsfs)))
        status = np.zeros(len(split), dtype=np.int32)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for i, (s, p) in enumerate(zip(split, params)):
                v, _, st, _ = oracle_eval(self.times, self.lh, self.bands, [], self.flags, 0, float(s), list(p), jsfs)
                llk[i], status[i] = v, st
        return SimpleNamespace(llk=llk, status=status)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, interleave, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from misti_amd.dist import evaluate_sharded, shard_indices
    split, params, jsfs = tiny_grid()[4:]
    eng = OracleEngine()
    out = evaluate_sharded(eng.evaluate, split, params, jsfs, interleave=interleave)
    mine = shard_indices(len(split), rank, world, interleave)
    q.put((rank, out.numpy(), eng.chains_seen, len(mine)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,interleave", [(2, True), (2, False), (3, True)])
def test_one_grid_sharded_over_ranks_equals_the_unsharded_evaluation(world, interleave):
    split, params, jsfs = tiny_grid()[4:]
    want = OracleEngine().evaluate(split, params, jsfs).llk
    assert np.isfinite(want).all()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, interleave, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    for rank, got, chains, n_mine in res:
        assert got.shape == want.shape
        assert np.array_equal(got, want), (rank, np.abs(got - want).max())     # same oracle, same candidates: the same bits
        assert n_mine in (len(split) // world, len(split) // world + 1)
        assert 1 <= chains <= 3                                                 # the grid's 3 chains span the ranks


def fake_eval(split, params, jsfs):
    s = np.asarray(split)[:, None]
    p = np.zeros_like(s) if params is None else np.asarray(params).sum(axis=1)[:, None]
    r = np.asarray(jsfs)[:, 1][None, :]
    return 1000.0 * s + 10.0 * p + 0.001 * r


def _worker_ragged(rank, world, port, n, interleave, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from misti_amd.dist import evaluate_sharded
    rng = np.random.default_rng(0)
    split = rng.integers(10, 90, n).astype(float)
    params = rng.random((n, 2))
    jsfs = rng.random((3, 8))
    out = evaluate_sharded(fake_eval, split, params, jsfs, interleave=interleave)
    q.put((rank, bool(np.array_equal(out.numpy(), fake_eval(split, params, jsfs)))))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,interleave", [(2, 37, True), (3, 10, True), (2, 1, True), (3, 2, False)])
def test_ragged_and_empty_shards(world, n, interleave):
    """Sizes that do not divide: the gather pads every rank to the same row count; a rank may own no candidate at all."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ragged, args=(r, world, port, n, interleave, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res


def test_shard_indices_partition():
    from misti_amd.dist import shard_indices
    for n in (0, 1, 7, 4096):
        for w in (1, 2, 3, 8):
            for il in (True, False):
                parts = [shard_indices(n, r, w, il) for r in range(w)]
                allidx = np.sort(np.concatenate(parts)) if parts else np.array([])
                assert np.array_equal(allidx, np.arange(n))


def _worker_chain(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from misti_amd.dist import chain_shards, evaluate_sharded
    split, params, jsfs = tiny_grid()[4:]
    eng = OracleEngine()
    out = evaluate_sharded(eng.evaluate, split, params, jsfs, by_chain=True)
    mine = chain_shards(params, len(split), world)[rank]
    q.put((rank, out.numpy(), eng.chains_seen, len(mine)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharding_by_chain_keeps_chains_whole_and_changes_no_value(world):
    """`by_chain=True`: whole chains (identical parameter vectors) are dealt to the ranks - the grid's 3 chains of 4 splits each
    land on distinct ranks, none is computed twice - and the gathered table equals the unsharded evaluation bit for bit."""
    split, params, jsfs = tiny_grid()[4:]
    want = OracleEngine().evaluate(split, params, jsfs).llk
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_chain, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sum(chains for _, _, chains, _ in res) == 3                            # every chain on exactly one rank
    assert sum(n for _, _, _, n in res) == len(split)
    for rank, got, chains, n_mine in res:
        assert np.array_equal(got, want), rank
        assert n_mine == 4 * chains                                               # all four splits of a chain travel with it


def test_chain_shards_partition_and_balance():
    from misti_amd.dist import chain_shards
    rng = np.random.default_rng(1)
    rates = rng.random((37, 2))
    params = np.repeat(rates, 5, axis=0)[rng.permutation(37 * 5)]                 # 37 chains x 5 members, shuffled
    for world in (1, 2, 3, 8):
        sh = chain_shards(params, len(params), world)
        assert np.array_equal(np.sort(np.concatenate(sh)), np.arange(len(params)))
        for s in sh:                                                               # a chain never spans ranks
            keys = {tuple(params[i]) for i in s}
            assert len(s) == 5 * len(keys)
        assert max(len(s) for s in sh) - min(len(s) for s in sh) <= 5
    assert [len(s) for s in chain_shards(None, 10, 3)] == [4, 3, 3]


def test_chains_are_dealt_by_cost_longest_first():
    """Round 5: chains cost their LENGTH (largest split index of their members) + 1/64 per member, and are dealt
    longest-processing-time-first - per-rank summed cost differs by less than one chain, where round-robin in order of first
    appearance (round 4) left a rank with the long chains up to 40 % more work on a grid ordered by split."""
    from misti_amd.dist import MEMBER_COST, chain_costs, chain_shards, deal_lpt
    rng = np.random.default_rng(7)
    n_chain = 61
    rates = rng.random((n_chain, 2))
    longest = rng.integers(20, 120, n_chain)                                       # every chain its own largest split
    split, params = [], []
    for c in range(n_chain):
        m = int(rng.integers(1, 9))
        st = np.sort(rng.integers(10, longest[c] + 1, m)).astype(float)
        st[-1] = longest[c] if c % 5 else longest[c] - 0.5                         # some chains end on a fractional split
        split += list(st)
        params += [rates[c]] * m
    order = rng.permutation(len(split))
    split, params = np.array(split)[order], np.array(params)[order]
    chain, cost = chain_costs(params, len(split), split)
    assert len(cost) == n_chain
    for c in range(n_chain):                                                       # cost = ceil(largest split) + members / 64
        mine = chain == c
        assert cost[c] == np.ceil(split[mine].max()) + MEMBER_COST * mine.sum()
        assert len({tuple(q) for q in params[mine]}) == 1
    for world in (2, 3, 4, 8):
        sh = chain_shards(params, len(split), world, split)
        assert np.array_equal(np.sort(np.concatenate(sh)), np.arange(len(split)))
        load = np.array([cost[np.unique(chain[s])].sum() for s in sh])
        for s in sh:                                                               # a chain never spans ranks
            assert len(s) == sum((chain == c).sum() for c in np.unique(chain[s]))
        assert load.max() - load.min() <= cost.max(), (world, load)                # "differs by at most one chain"
        assert load.max() <= cost.sum() / world + cost.max()
        rr = np.zeros(world)                                                       # round 4's deal of the same grid
        for c in range(n_chain):
            rr[c % world] += cost[c]
        assert load.max() <= rr.max() + 1e-9
    # equal costs: round-robin in order of first appearance, as before
    assert list(deal_lpt(np.ones(7), 3)) == [0, 1, 2, 0, 1, 2, 0]
    # band bounds are part of the chain key
    bb = np.zeros((len(split), 1, 2), dtype=np.int32)
    bb[::2, 0, 0] = 3
    assert len(chain_costs(params, len(split), split, bb)[1]) > n_chain


def _worker_status(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from misti_amd.dist import evaluate_sharded
    split, params, jsfs = tiny_grid()[4:]
    eng = OracleEngine()
    llk, status = evaluate_sharded(eng.evaluate, split, params, jsfs, by_chain=True, with_status=True)
    q.put((rank, llk.numpy(), status.numpy(), eng.chains_seen))
    dist.destroy_process_group()


def test_more_ranks_than_chains_with_status():
    """`cli --gpus N` gathers [llk | status] with whole chains per rank; with more ranks than chains (3 chains,
    4 ranks) one rank has no candidate - it used to die in `reshape(0, -1)` before the collective and take the sweep with it."""
    world = 4
    split, params, jsfs = tiny_grid()[4:]
    want = OracleEngine().evaluate(split, params, jsfs)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_status, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(ch for _, _, _, ch in res if ch is not None) == [1, 1, 1] and sum(ch is None for _, _, _, ch in res) == 1
    for rank, llk, status, _ in res:
        assert np.array_equal(llk, want.llk) and np.array_equal(status, want.status) and status.dtype == np.int32, rank


def _worker_few_starts(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from misti_amd.optimize import solve_batched_dev
    starts, _ = _search_inputs()
    x, llh, r = solve_batched_dev(StubSearchEngine(), 20.0, starts[:2], np.ones(8), tol=1e-6)
    q.put((rank, x, llh, r["nfev"]))
    dist.destroy_process_group()


def test_fewer_starts_than_ranks():
    """two starts over three ranks - the rank with an empty block takes part in the gather."""
    world = 3
    starts, _ = _search_inputs()
    want = StubSearchEngine().nm_solve(starts[:2], 20.0, None, tol=1e-6)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_few_starts, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    for rank, x, llh, nfev in res:
        assert np.array_equal(x, want["x"]) and np.array_equal(llh, want["llh"]) and np.array_equal(nfev, want["nfev"]), rank


# ---- round 4: the product entry points that use more than one GPU, rehearsed over gloo -----------------------------------------
def _cli_fixture(tmp_path):
    import json
    from conftest import GOLDEN
    cli = json.load(open(os.path.join(GOLDEN, "golden_host.json")))["cli"]
    for name, key in (("g1.psmc", "psmc1"), ("g2.psmc", "psmc2"), ("data.sfs", "jsfs"), ("setunits.txt", "units")):
        (tmp_path / name).write_text(cli[key])
    return ["g1.psmc", "g2.psmc", "data.sfs", "20", "-wd", str(tmp_path), "--funits", str(tmp_path / "setunits.txt"), "--cpfit",
            "-mi", "1", "2", "20", "0.1", "1", "--grid-st", "18", "21", "--grid-mi", "0", "0.05", "0.4", "3", "--all-bs"]


def test_cli_gpus_starts_its_own_ranks_and_prints_the_unsharded_sweep(tmp_path):
    """`python -m misti_amd.cli ... --grid-st --grid-mi --all-bs --gpus 2`: the process starts two ranks (torch.distributed.run over
    gloo here, RCCL on a node), whole chains are dealt to them, rank 0 prints - the same lines as the one-process sweep.  The oracle
    stands in for the engine (tests/cli_oracle_hook.py wraps the module from the outside: no GPU here); the parent never imports torch."""
    import subprocess
    args = _cli_fixture(tmp_path)
    env = dict(os.environ, MISTI_DIST_BACKEND="gloo", PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]))
    one = subprocess.run([sys.executable, "-m", "cli_oracle_hook"] + args, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-X", "importtime", "-m", "cli_oracle_hook"] + args + ["--gpus", "2"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    rows = lambda text: [l for l in text.splitlines() if l.startswith("bs_id =") or l.startswith("best:") or l.startswith("bootstrap:")]
    assert len(rows(one.stdout)) == 4 * 3 * 5 + 2 and rows(two.stdout) == rows(one.stdout)
    assert "Sharded over 2 ranks" in two.stdout and "Sharded" not in one.stdout
    parent_imports = [l for l in two.stderr.splitlines() if l.startswith("import time:")]
    assert parent_imports and not any(l.rstrip().endswith(" torch") for l in parent_imports)      # the launching process stays off torch / HIP


def test_cli_gpus_without_a_gpu_fails_loudly(tmp_path):
    """No CPU path behind --gpus either: the product module itself, without a GPU, fails on every rank."""
    import subprocess
    args = _cli_fixture(tmp_path)
    env = dict(os.environ, MISTI_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "misti_amd.cli"] + args + ["--gpus", "2"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode != 0 and "bs_id =" not in r.stdout


class StubSearchEngine:
    """Engine-shaped object whose nm_solve / basinhopping are SciPy itself on an analytic objective: what is rehearsed is the
    dealing of starts (and of their generators) to the ranks and the gather, not the minimiser."""
    n_param = 2

    @staticmethod
    def f(x):
        return (x[0] - 0.3) ** 2 + 3.0 * (x[1] - 0.1) ** 2 + 0.1 * np.sin(8 * x[0]) * np.cos(5 * x[1])

    def nm_solve(self, starts, split_time, jsfs_row, tol=1e-4, maxiter=1000):
        from scipy import optimize
        rs = [optimize.minimize(self.f, s, method="Nelder-Mead", options=dict(xatol=tol, fatol=tol, maxiter=maxiter)) for s in starts]
        return dict(x=np.array([r.x for r in rs]).reshape(len(rs), 2), llh=np.array([-r.fun for r in rs]), nit=np.array([r.nit for r in rs], dtype=np.int32),
                    nfev=np.array([r.nfev for r in rs], dtype=np.int32), status=np.array([r.status for r in rs], dtype=np.int32), iterations_issued=7)

    def basinhopping(self, starts, split_time, jsfs_row, rngs, niter=3, **kw):
        from scipy import optimize
        rs = [optimize.basinhopping(self.f, s, niter=niter, T=0.5, stepsize=0.05, minimizer_kwargs=dict(method="Nelder-Mead"), rng=np.random.default_rng(g))
              for s, g in zip(starts, rngs)]
        return dict(x=np.array([r.x for r in rs]).reshape(len(rs), 2), llh=np.array([-r.fun for r in rs]), nfev=np.array([r.nfev for r in rs], dtype=np.int32),
                    failures=np.array([r.minimization_failures for r in rs], dtype=np.int32), accepted=np.zeros(len(rs), dtype=np.int32))


def _search_inputs():
    rng = np.random.default_rng(3)
    return rng.uniform(0.0, 1.0, (7, 2)), [900 + s for s in range(7)]


def _worker_search(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from misti_amd.dist import best_per_replicate, bootstrap_sharded
    from misti_amd.optimize import basinhopping_dev, solve_batched_dev
    starts, seeds = _search_inputs()
    eng = StubSearchEngine()
    x, llh, r = solve_batched_dev(eng, 20.0, starts, np.ones(8), tol=1e-6)
    bh = basinhopping_dev(eng, 20.0, starts, np.ones(8), seeds, niter=3)
    split, params, jsfs = tiny_grid()[4:]
    rows = np.vstack([jsfs, jsfs * 0.5, jsfs * 2.0, jsfs[:1] * 0.25])           # 7 replicates over the ranks
    table = fake_eval(split, params, rows)
    best = bootstrap_sharded(lambda a, b: best_per_replicate(table[:, a:b]), rows.shape[0])
    q.put((rank, x, llh, {k: r[k] for k in ("nit", "nfev", "status")}, bh, best))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_searches_and_bootstrap_scan_sharded_over_ranks(world):
    """optimize.solve_batched_dev / basinhopping_dev / the bootstrap scan inside a process group: starts (with their generators) and
    replicates are dealt to the ranks in contiguous blocks, one all_gather each; every rank ends with the unsharded result."""
    from misti_amd.dist import best_per_replicate
    starts, seeds = _search_inputs()
    eng = StubSearchEngine()
    want = eng.nm_solve(starts, 20.0, None, tol=1e-6)
    want_bh = eng.basinhopping(starts, 20.0, None, seeds, niter=3)
    split, params, jsfs = tiny_grid()[4:]
    rows = np.vstack([jsfs, jsfs * 0.5, jsfs * 2.0, jsfs[:1] * 0.25])
    want_best = best_per_replicate(fake_eval(split, params, rows))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_search, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
    for rank, x, llh, counters, bh, best in res:
        assert np.array_equal(x, want["x"]) and np.array_equal(llh, want["llh"]), rank
        for k in ("nit", "nfev", "status"):
            assert np.array_equal(counters[k], want[k]) and counters[k].dtype == np.int32, (rank, k)
        for k in ("x", "llh", "nfev", "failures"):
            assert np.array_equal(bh[k], want_bh[k]), (rank, k)
        assert np.array_equal(best, want_best), rank


def test_block_bounds_partition():
    from misti_amd.dist import block_bounds
    for n in (0, 1, 7, 1000, 16384):
        for w in (1, 2, 3, 8):
            lo = block_bounds(n, w)
            assert lo[0] == 0 and lo[-1] == n and all(0 <= lo[r + 1] - lo[r] <= -(-n // w) for r in range(w))
