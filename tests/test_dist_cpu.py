"""Multi-rank path on CPU: world_size 2 (and 3) gloo processes shard a candidate batch and
gather the log-likelihoods; the evaluator is a stand-in closed form so no GPU is needed."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def fake_eval(split, params, jsfs):
    s = np.asarray(split)[:, None]
    p = np.zeros_like(s) if params is None else np.asarray(params).sum(axis=1)[:, None]
    r = np.asarray(jsfs)[:, 1][None, :]
    return 1000.0 * s + 10.0 * p + 0.001 * r


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, interleave, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from misti_amd.dist import evaluate_sharded, shard_indices
    rng = np.random.default_rng(0)
    split = rng.integers(10, 90, n).astype(float)
    params = rng.random((n, 2))
    jsfs = rng.random((3, 8))
    out = evaluate_sharded(fake_eval, split, params, jsfs, interleave=interleave)
    want = fake_eval(split, params, jsfs)
    ok = bool(np.array_equal(out.numpy(), want)) and len(shard_indices(n, rank, world, interleave)) in (n // world, n // world + 1)
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,interleave", [(2, 37, True), (2, 64, False), (3, 10, True), (2, 1, True)])
def test_sharded_gather(world, n, interleave):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, interleave, q)) for r in range(world)]
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
