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
