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


def evaluate_sharded(evaluate, split_time, params, jsfs, interleave=True, group=None, device=None):
    """Shard candidates over the ranks of the default process group and gather ``llk``.

    ``evaluate(split[n_loc], params[n_loc, P] or None, jsfs[R, 8])`` is the per-rank evaluator: it returns
    ``llk[n_loc, R]`` (a NumPy array or a torch tensor) or an object with an ``llk`` attribute - on the GPU box
    ``Engine.evaluate`` (a ``BatchResult``); the oracle in the CPU tests.  With the nccl backend (RCCL) the
    gathered tensor must live on the rank's GPU: ``device`` defaults to the current CUDA device there.
    Returns ``llk[n_total, R]`` in candidate order on every rank.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    split_time = np.asarray(split_time, dtype=np.float64)
    n = split_time.shape[0]
    idx = shard_indices(n, rank, world, interleave)
    p_loc = None if params is None else np.asarray(params, dtype=np.float64)[idx]
    llk = evaluate(split_time[idx], p_loc, jsfs)
    llk = getattr(llk, "llk", llk)                       # a BatchResult
    llk = torch.as_tensor(llk, dtype=torch.float64)
    if device is None and dist.is_initialized() and dist.get_backend(group) == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    if device is not None:
        llk = llk.to(device)
    return gather_rows(llk, n, rank, world, interleave, group)
