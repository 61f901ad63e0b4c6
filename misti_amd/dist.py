"""Multi-GPU sharding of a candidate batch: one process per GPU, no data-path
exchange during evaluation, one gather of log-likelihoods at the end.

The reference has no counterpart: it fans a grid out over OS processes with GNU
parallel and concatenates stdout (``README.md:110-115``, ``test.bs/*.sh``).
Here candidates are independent, so rank r evaluates its own contiguous or
interleaved share and the ``[n_cand_local, n_rep]`` blocks are gathered with
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  Message sizes are tiny (tens to hundreds of KB), so one
collective per batch on the compute stream is enough.
"""
from __future__ import annotations

import os

import numpy as np


def env_rank():
    """(rank, local_rank, world_size) from the torchrun environment (defaults 0, 0, 1)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_indices(n, rank, world, interleave=True):
    """Indices of the candidates rank ``rank`` evaluates.

    Interleaving (round-robin) balances load when the grid is ordered by split
    index: cost grows with the number of two-population intervals."""
    if interleave:
        return np.arange(rank, n, world, dtype=np.int64)
    per = -(-n // world)
    return np.arange(min(n, rank * per), min(n, (rank + 1) * per), dtype=np.int64)


def padded_count(n, world):
    return -(-n // world)


def chain_shards(params, n, world):
    """Candidate indices per rank when whole CHAINS are dealt out (``[shard_0, ..., shard_{world-1}]``).

    Candidates with identical parameter vectors share one lambda-correction chain, computed once per rank
    that holds any of them, and a chain costs its full latency however few members a rank has: interleaving a split x rate
    grid puts every chain on every rank - kernel 1 is then not sharded at all (BASELINE config 5: 2 048 chains x 32 members;
    eight interleaved ranks each still run 2 048 chains).  Dealing chains keeps a chain on ONE rank, all its splits with it
    (so ranks stay balanced): 256 chains per rank on eight GPUs - the one-chain-per-wave path.  Chains are dealt round-robin in
    order of first appearance; without parameters (``params is None``: one chain) the split of ``shard_indices`` is used."""
    if params is None or world == 1:
        return [shard_indices(n, r, world, interleave=True) for r in range(world)]
    p = np.ascontiguousarray(np.asarray(params, dtype=np.float64).reshape(n, -1))
    _, first, inverse = np.unique(p.view(np.dtype((np.void, p.dtype.itemsize * p.shape[1]))).ravel(), return_index=True, return_inverse=True)
    rank_of_chain = np.empty(len(first), dtype=np.int64)
    rank_of_chain[np.argsort(first, kind="stable")] = np.arange(len(first)) % world      # round-robin in order of first appearance
    owner = rank_of_chain[inverse.ravel()]
    return [np.nonzero(owner == r)[0].astype(np.int64) for r in range(world)]


def gather_shards(local, shards, rank, world, group=None):
    """All-gather row blocks of UNEQUAL, explicitly listed shards back into candidate order (see ``chain_shards``)."""
    import torch
    import torch.distributed as dist

    n_total = int(sum(len(s) for s in shards))
    per = max((len(s) for s in shards), default=0)
    pad = torch.full((per,) + tuple(local.shape[1:]), float("nan"), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if world == 1:
        parts = pad.unsqueeze(0)
    else:
        flat = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(flat, pad, group=group)
        parts = flat.view((world, per) + tuple(local.shape[1:]))
    out = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = torch.as_tensor(shards[r], device=local.device)
        out[idx] = parts[r, : idx.numel()]
    return out


def gather_rows(local, n_total, rank, world, interleave=True, group=None):
    """All-gather per-rank row blocks back into candidate order.

    ``local`` is a torch tensor ``[n_local, ...]`` holding the rows of
    ``shard_indices(n_total, rank, world)``.  Returns ``[n_total, ...]`` on every rank.
    """
    import torch
    import torch.distributed as dist

    per = padded_count(n_total, world)
    pad = torch.full((per,) + tuple(local.shape[1:]), float("nan"), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if world == 1:
        parts = pad.unsqueeze(0)
    else:
        flat = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(flat, pad, group=group)
        parts = flat.view((world, per) + tuple(local.shape[1:]))
    out = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = torch.as_tensor(shard_indices(n_total, r, world, interleave), device=local.device)
        out[idx] = parts[r, : idx.numel()]
    return out


def evaluate_sharded(evaluate, split_time, params, jsfs, interleave=True, group=None, device=None, by_chain=False):
    """Shard candidates over the ranks of the default process group and gather ``llk``.

    ``evaluate(split[n_loc], params[n_loc, P] or None, jsfs[R, 8])`` is the per-rank evaluator: it returns
    ``llk[n_loc, R]`` (a NumPy array or a torch tensor) or an object with an ``llk`` attribute - on the GPU box
    ``Engine.evaluate`` (a ``BatchResult``); the oracle in the CPU tests.  With the nccl backend (RCCL) the
    gathered tensor must live on the rank's GPU: ``device`` defaults to the current CUDA device there.
    ``by_chain=True`` deals whole chains to the ranks instead of interleaving candidates (``chain_shards``).
    Returns ``llk[n_total, R]`` in candidate order on every rank.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    split_time = np.asarray(split_time, dtype=np.float64)
    n = split_time.shape[0]
    shards = chain_shards(params, n, world) if by_chain else None      # whole chains per rank (see chain_shards)
    idx = shards[rank] if by_chain else shard_indices(n, rank, world, interleave)
    p_loc = None if params is None else np.asarray(params, dtype=np.float64)[idx]
    llk = evaluate(split_time[idx], p_loc, jsfs)
    llk = getattr(llk, "llk", llk)                       # a BatchResult
    llk = torch.as_tensor(llk, dtype=torch.float64)
    if device is None and dist.is_initialized() and dist.get_backend(group) == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    if device is not None:
        llk = llk.to(device)
    if by_chain:
        return gather_shards(llk, shards, rank, world, group)
    return gather_rows(llk, n, rank, world, interleave, group)
