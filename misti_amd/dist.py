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


MEMBER_COST = 1.0 / 64.0      # cost of one member of a chain (spectrum kernel) in units of one corrected interval (misti_multi.cpp)


def chain_costs(params, n, split_time=None, band_bounds=None, numT=None):
    """Chains of a batch and what each costs.  Candidates with bitwise identical parameter vectors (and band bounds) share one
    lambda-correction chain; a chain costs its LENGTH - the corrected two-population intervals up to the largest split index of
    its members, ``ceil(split)``: a fractional split adds its shortened interval - plus ``MEMBER_COST`` per member.  Without
    ``split_time`` every chain costs the same.  Returns ``(chain_of_candidate[n], cost[n_chain])``, chains numbered in order of
    first appearance."""
    p = np.ascontiguousarray(np.asarray(params, dtype=np.float64).reshape(n, -1))
    key = p.view(np.uint8).reshape(n, -1)
    if band_bounds is not None:
        key = np.concatenate([key, np.ascontiguousarray(band_bounds, dtype=np.int32).reshape(n, -1).view(np.uint8).reshape(n, -1)], axis=1)
    key = np.ascontiguousarray(key)
    _, first, inverse = np.unique(key.view(np.dtype((np.void, key.shape[1]))).ravel(), return_index=True, return_inverse=True)
    rank_by_first = np.empty(len(first), dtype=np.int64)
    rank_by_first[np.argsort(first, kind="stable")] = np.arange(len(first))
    chain = rank_by_first[inverse.ravel()]
    members = np.bincount(chain, minlength=len(first)).astype(np.float64)
    length = np.zeros(len(first))
    if split_time is not None:
        st = np.asarray(split_time, dtype=np.float64).reshape(n)
        ln = np.where((st == st) & (st > 0), np.ceil(np.minimum(st, np.inf if numT is None else float(numT))), 0.0)
        np.maximum.at(length, chain, ln)
    return chain, length + MEMBER_COST * members


def deal_lpt(cost, world):
    """Longest-processing-time-first: items by descending cost (ties: lowest index), each to the bin with the least cost so far
    (ties: lowest bin).  Returns ``owner[len(cost)]``.  Within 4/3 of the optimal makespan; equal costs deal round-robin."""
    load = np.zeros(world)
    owner = np.empty(len(cost), dtype=np.int64)
    for i in np.argsort(-np.asarray(cost, dtype=np.float64), kind="stable"):
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += cost[i]
    return owner


def chain_shards(params, n, world, split_time=None, band_bounds=None, numT=None):
    """Candidate indices per rank when whole CHAINS are dealt out (``[shard_0, ..., shard_{world-1}]``).

    Candidates with identical parameter vectors share one lambda-correction chain, computed once per rank
    that holds any of them, and a chain costs its full latency however few members a rank has: interleaving a split x rate
    grid puts every chain on every rank - kernel 1 is then not sharded at all (BASELINE config 5: 2 048 chains x 32 members;
    eight interleaved ranks each still run 2 048 chains).  Dealing chains keeps a chain on ONE rank, all its splits with it:
    256 chains per rank on eight GPUs - the one-chain-per-wave path.  Chains are dealt by COST, the longest first, each to the
    rank with the least work so far (``chain_costs``, ``deal_lpt``; ``misti_multi_eval_batch`` deals its contexts the same way):
    per-rank summed cost differs by less than one chain.  Without ``split_time`` all chains cost the same and the deal is
    round-robin in order of first appearance; without parameters (``params is None``: one chain) the split of ``shard_indices``
    is used."""
    if params is None or world == 1:
        return [shard_indices(n, r, world, interleave=True) for r in range(world)]
    chain, cost = chain_costs(params, n, split_time, band_bounds, numT)
    owner = deal_lpt(cost, world)[chain]
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


def init_from_env(backend=None):
    """Join the process group a launcher (``launch_ranks``, torchrun) prepared: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment.  Backend: ``nccl`` (= RCCL over xGMI) when this rank has a GPU, else ``gloo`` (the CPU tests); the rank's GPU is
    LOCAL_RANK.  Returns (rank, local_rank, world); a no-op returning (0, 0, 1) without WORLD_SIZE."""
    rank, local, world = env_rank()
    if "WORLD_SIZE" not in os.environ:
        return 0, 0, 1
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("MISTI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def launch_ranks(n, argv, module=None, script=None, port=0, cwd=None):
    """Start ``n`` ranks of ``python -m <module> argv`` (or ``python <script> argv``) as CHILD processes with
    ``torch.distributed.run`` (one per GPU, rendezvous on 127.0.0.1) and return (exit code, stdout of the ranks).  The calling
    process must not have touched the GPU and never does: a process that has initialised HIP must not be forked or replaced.
    The reference's counterpart is ``parallel -j N`` over OS processes (``/root/reference/README.md:110-115``)."""
    import socket
    import subprocess
    import sys
    if not port:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
           "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += ["-m", module] if module else [script]
    cmd += list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=sys.stderr, cwd=cwd)
    return r.returncode, r.stdout.decode(errors="replace")


def block_bounds(n, world):
    """Contiguous blocks of starts / replicates: rank r owns [lo[r], lo[r + 1]) (as misti_multi_nm_solve deals them)."""
    return [n * r // world for r in range(world + 1)]


def gather_blocks(local, n_total, rank, world, group=None):
    """All-gather the row blocks of ``block_bounds`` (unequal by at most one row) back into order: ``[n_total, ...]`` on every rank."""
    import torch
    import torch.distributed as dist
    lo = block_bounds(n_total, world)
    if world == 1:
        return local
    per = max(lo[r + 1] - lo[r] for r in range(world))
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    flat = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(flat, pad, group=group)
    parts = flat.view((world, per) + tuple(local.shape[1:]))
    return torch.cat([parts[r, : lo[r + 1] - lo[r]] for r in range(world)], dim=0)


def _group_state(group=None):
    import torch.distributed as dist
    if not dist.is_initialized():
        return 0, 1, None
    return dist.get_rank(group), dist.get_world_size(group), dist.get_backend(group)


def search_sharded(search, starts, keys, group=None, device=None, **per_start):
    """Independent searches (Nelder-Mead or basin-hopping starts: BASELINE config 3) over the ranks of the process group: rank r
    runs ``search(starts[block r], **{k: v[block r]})`` - ``Engine.nm_solve`` / ``Engine.basinhopping`` bound to their other
    arguments; ``per_start`` holds arguments indexed by start (basin hopping's ``rngs``) - and ONE all_gather returns, on every rank,
    the dict of per-start arrays ``keys`` in start order.  A start's trajectory does not depend on what else travels in its batches,
    so the result equals the single-device call's.  Scalars of the per-rank result (work counters) are dropped."""
    import torch
    rank, world, backend = _group_state(group)
    starts = np.atleast_2d(np.asarray(starts, dtype=np.float64))
    S = starts.shape[0]
    lo = block_bounds(S, world)
    a, b = lo[rank], lo[rank + 1]
    kw = {k: v[a:b] for k, v in per_start.items()}
    res = search(starts[a:b], **kw) if b > a else {k: np.zeros((0,) + ((starts.shape[1],) if k == "x" else ())) for k in keys}
    if world == 1:
        return {k: np.asarray(res[k]) for k in keys}
    if device is None and backend == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    cols, widths = [], []
    for k in keys:                                     # one collective for all keys: a float64 table [block, sum of widths]
        v = np.asarray(res[k], dtype=np.float64).reshape(b - a, starts.shape[1] if k == "x" else 1)
        cols.append(v)
        widths.append(v.shape[1])
    table = torch.as_tensor(np.concatenate(cols, axis=1) if cols else np.zeros((b - a, 0)))
    if device is not None:
        table = table.to(device)
    full = gather_blocks(table, S, rank, world, group).cpu().numpy()
    out, c = {}, 0
    for k, wd in zip(keys, widths):
        v = full[:, c:c + wd]
        c += wd
        out[k] = v if k == "x" else v[:, 0]
        if k in ("nit", "nfev", "status", "failures", "accepted"):
            out[k] = out[k].astype(np.int32)
    return out


def best_per_replicate(llk):
    """Per column (replicate) of ``llk[n_cand][n_rep]`` the row with the largest value; -inf and NaN never win, ties go to the
    lowest index, -1 where no candidate has a value (what ``misti_argmax_dev`` computes on the device)."""
    llk = getattr(llk, "llk", llk)
    llk = np.asarray(llk.cpu() if hasattr(llk, "cpu") else llk, dtype=np.float64)
    masked = np.where(np.isfinite(llk), llk, -np.inf)
    return np.where(np.isfinite(masked.max(axis=0)), masked.argmax(axis=0), -1).astype(np.int64)


def bootstrap_sharded(best_of_block, n_rep, group=None, device=None):
    """A bootstrap scan (BASELINE config 4: split scan x 1 000 replicates) over the ranks: the REPLICATES are dealt out in contiguous
    blocks - rank r scans every split value against its block of JSFS rows (the spectra are recomputed per rank: they are one
    chain and cheap; the replicate epilogue and the arg-max are what is shared out) - and one all_gather of the per-replicate
    winning index returns ``best[n_rep]`` on every rank.  ``best_of_block(a, b)`` -> winning candidate index for replicates
    a .. b-1 (``optimize._scan_best`` on the rank's GPU; ``best_per_replicate`` of a host table in the CPU tests)."""
    import torch
    rank, world, backend = _group_state(group)
    lo = block_bounds(int(n_rep), world)
    a, b = lo[rank], lo[rank + 1]
    best = np.asarray(best_of_block(a, b), dtype=np.int64) if b > a else np.zeros(0, dtype=np.int64)
    if world == 1:
        return best
    if device is None and backend == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.as_tensor(best)
    if device is not None:
        t = t.to(device)
    return gather_blocks(t, int(n_rep), rank, world, group).cpu().numpy()


def evaluate_sharded(evaluate, split_time, params, jsfs, interleave=True, group=None, device=None, by_chain=False, with_status=False):
    """Shard candidates over the ranks of the default process group and gather ``llk``.

    ``evaluate(split[n_loc], params[n_loc, P] or None, jsfs[R, 8])`` is the per-rank evaluator: it returns
    ``llk[n_loc, R]`` (a NumPy array or a torch tensor) or an object with an ``llk`` attribute - on the GPU box
    ``Engine.evaluate`` (a ``BatchResult``); the oracle in the CPU tests.  With the nccl backend (RCCL) the
    gathered tensor must live on the rank's GPU: ``device`` defaults to the current CUDA device there.
    ``by_chain=True`` deals whole chains to the ranks instead of interleaving candidates (``chain_shards``).
    Returns ``llk[n_total, R]`` in candidate order on every rank; with ``with_status=True`` (the evaluator's result must carry
    ``status``) the per-candidate status travels in the same collective and ``(llk, status[n_total])`` is returned.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    split_time = np.asarray(split_time, dtype=np.float64)
    n = split_time.shape[0]
    shards = chain_shards(params, n, world, split_time) if by_chain else None      # whole chains per rank, dealt by cost (see chain_shards)
    idx = shards[rank] if by_chain else shard_indices(n, rank, world, interleave)
    p_loc = None if params is None else np.asarray(params, dtype=np.float64)[idx]
    n_rep = 0 if jsfs is None else int(np.asarray(jsfs).reshape(-1, 8).shape[0])
    if len(idx):
        res = evaluate(split_time[idx], p_loc, jsfs)
        llk = getattr(res, "llk", res)                   # a BatchResult
        llk = torch.as_tensor(llk, dtype=torch.float64).reshape(len(idx), n_rep)
        st = torch.as_tensor(np.asarray(res.status), dtype=torch.float64).reshape(len(idx), 1) if with_status else None
    else:
        # a rank without a candidate (more ranks than chains): it still takes part in the collective, with an empty block of
        # the right width - reshape(0, -1) is ambiguous and the rank would die before the all_gather
        llk = torch.empty((0, n_rep), dtype=torch.float64)
        st = torch.empty((0, 1), dtype=torch.float64) if with_status else None
    if with_status:                                      # one collective: the status rides as an extra column
        llk = torch.cat([llk, st], dim=1)
    if device is None and dist.is_initialized() and dist.get_backend(group) == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    if device is not None:
        llk = llk.to(device)
    out = gather_shards(llk, shards, rank, world, group) if by_chain else gather_rows(llk, n, rank, world, interleave, group)
    if with_status:
        return out[:, :-1], out[:, -1].to(torch.int32)
    return out
