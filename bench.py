#!/usr/bin/env python3
"""Headline benchmark: composite-llk evaluations/s over a (split x mi-rate) grid at
128 merged PSMC intervals (BASELINE.json `metric`, configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Started WITHOUT torchrun and with --gpus N > 1 (no WORLD_SIZE in the environment) this process never touches the
GPU: it starts the second command line as a child (one rank per GPU), forwards rank 0's single JSON line and
exits with the child's code (`launch_ranks`).

A step = one pass of the hot path (prepare, chain discovery, lambda-correction of the chains
with their trunks following, tails, spectrum kernel with the replicate epilogue) over one
batch: the 4 096-point grid of config 2 (64 split indices x 64 rates of one band
`-mi 1 4 {st} {r} 1`, `--cpfit`, numT = 128) with inputs already resident in HBM.

What the line says ("make the bench line say what it measures"):
  value / ms_per_step      whole-job rate with `config.batches_in_flight` independent batches overlapped on as many
                           engine contexts + HIP streams (a batch alone is latency-bound by its longest
                           lambda-correction chain); W warmup steps, then EXACTLY K steps between barriers, repeated
                           until the timed regions add up to >= --min-seconds, median repetition reported
  single_batch             the same step issued strictly one after another on ONE stream: the per-grid latency and
                           rate (this is the figure to compare with "one grid sweep" as BASELINE.json words config 2)
  roofline                 dominant kernel, HIP events on its launch stream during the single-batch pass
  cpu_baseline             the NumPy/SciPy oracle on a bounded sample, in a CHILD PROCESS that never touched the GPU

--scaling weak (default): every rank its own grid, one all_gather of llk per batch (RCCL).
--scaling strong: ONE grid (use --workload config4 | config5, the 8-GPU configurations of BASELINE.json) sharded
over the ranks with misti_amd.dist.shard_indices (interleaved), one all_gather of llk per batch; the JSON carries
the process group's world size and the chain count of every rank.
Rank 0 prints ONE JSON line on stdout (everything else goes to stderr).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# Lanes (HIP streams with independent batches in flight).  Measured on MI355X / ROCm 7.2: own streams map
# one-to-one onto hardware queues; the device runs 23 of them beside each other and time-slices from the 24th
# ACTIVE one on (a burst of twenty batches: 10 ms instead of 2.7 - profiles/r06_hw_queue_cliff.txt), so the runtime
# is capped at 22 (HW_QUEUES; streams beyond the cap share queues: slower, never the cliff).  Defaults leave room for
# the null stream and, with several ranks, for RCCL's and torch's own streams.
# (single rank with RCCL initialised and one all_gather per 8 batches of a lane: 16 lanes 2.76e7, 18 lanes 2.86e7,
# 20 lanes 2.65e7, 22 lanes 2.80e7 evals/s against 2.97e7 without RCCL -> 18 with several ranks)
# (round 5, the driver's shape --steps 20 --warmup 5 with RCCL initialised on one rank: 18 lanes 2.38e7, 19 lanes 2.40e7, 20 lanes 0.80e7 with 24 or
# more queues - RCCL and c10d take the queues between 19 and 24, and the 24th is the cliff; under the cap of 22 (round 6): 18 lanes 2.31e7, 19 lanes
# 1.91e7, 20 lanes 1.86e7, nothing collapses.  ONE collective for all lanes' batches instead of one per lane was tried and is slower, 3.18 against
# 3.71e7 steady on 18 lanes: every lane then waits for the same collective)
# Several ranks: 17 lanes.  18 fit under the cap on ONE rank with RCCL initialised (18 lanes + the null stream + c10d's communicator stream + two
# more, RCCL's own = 22), but with a 19th stream the lanes start sharing queues (2.31e7 -> 1.91e7 in the driver's shape), and RCCL between several
# ranks has never run on this code: 17 lanes leave one queue spare and cost nothing in the driver's shape (20 steps: 2.31e7 with 17 as with 18
# lanes - two or three lanes two batches deep take the same time; 16 lanes 2.22e7).
DEFAULT_STREAMS = 20 if (int(os.environ.get("WORLD_SIZE", "1")) <= 1 and "--force-dist" not in sys.argv) else 17
HW_QUEUES = 22                 # the cap misti_lanes.cpp sets (a 24th ACTIVE queue is a cliff: profiles/r06_hw_queue_cliff.txt); --hw-queues overrides it


def _early_streams():
    """--streams must reach the HIP runtime (hardware queue count) before it initialises."""
    n, q = DEFAULT_STREAMS, None
    for i, a in enumerate(sys.argv):
        for name in ("--streams", "--hw-queues"):
            v = None
            if a == name and i + 1 < len(sys.argv):
                v = int(sys.argv[i + 1])
            elif a.startswith(name + "="):
                v = int(a.split("=", 1)[1])
            if v is not None:
                n, q = (v, q) if name == "--streams" else (n, v)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(q if q else HW_QUEUES))      # more lanes than queues share queues; never more queues than the cap


_early_streams()

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak
GPU_CLOCK_HZ = 2.4e9           # MI355X peak engine clock (MI355X_MICROARCH.md)
N_SIMD = 1024                  # 256 CUs x 4 SIMDs


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="config2",
                    help="config2 (headline) | config2x16 | config3 | config4 | config5 | config3-search | config3-basinhopping")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak: every rank its own grid; strong: ONE grid sharded over the ranks (config4 / config5)")
    ap.add_argument("--shard", default="chain", choices=("chain", "interleave"),
                    help="strong scaling: deal whole lambda-correction chains to the ranks (default; a chain interleaved over the ranks is recomputed "
                         "on every one of them) or interleave candidates (SURVEY 8e)")
    ap.add_argument("--pretend-world", type=int, default=0,
                    help="single GPU, --scaling strong: evaluate the shard rank 0 of an N-rank run would own (a per-rank cost model, not a scaling "
                         "curve; the line says so in config.pretend_world and counts only that shard's candidates)")
    ap.add_argument("--min-seconds", type=float, default=0.25, help="repeat the K-step timed loop until the timed regions add up to this; the median is reported")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample (wall)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fit", default="", choices=("", "cpfit", "default"),
                    help="override the workload's fitting mode: cpfit (--cpfit) or default (the reference's default: expected coalescence time)")
    ap.add_argument("--bh-niter", type=int, default=0, help="config3-basinhopping: hops per start (default: SciPy's 100, as the reference's Solve(globalOpt=True))")
    ap.add_argument("--bh-starts", type=int, default=0, help="config3-search / config3-basinhopping: number of random starts (default: 16 384, BASELINE config 3)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the single_call / host_abi / strong blocks (headline leg only)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the all_gather path even with one rank (rehearsal)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal without a GPU (tests/test_bench_launcher_cpu.py): ranks rendezvous over gloo, shard a stub grid with "
                         "misti_amd.dist and gather it; no engine, no measurement - the line says \"data\": \"dry-run\"")
    ap.add_argument("--gather-bucket", type=int, default=8,
                    help="multi-GPU: batches of a lane whose llk share ONE all_gather (fewer, larger collectives: xGMI is latency-bound at 32 KB)")
    ap.add_argument("--flush-mode", choices=["filled", "each"], default="filled",
                    help="multi-GPU: how the partly filled buckets are gathered at the end of a region (flush_all): per lane, filled slots only, "
                         "shallow lanes first | per lane, whole buckets, lane order (until round 6)")
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES (default: %d)" % HW_QUEUES)
    ap.add_argument("--streams", type=int, default=DEFAULT_STREAMS,
                    help="HIP streams the steps are issued on round-robin: independent batches overlap (1 = strictly serial)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when this process starts the ranks itself (default: a free one)")
    return ap.parse_args(argv)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a):
    """`python bench.py --gpus N` as the driver invokes it: start N ranks as CHILD processes (torch.distributed.run, one per
    GPU, rendezvous on 127.0.0.1) and forward rank 0's JSON line.  This process has not imported torch or touched HIP and
    never does - a process that has initialised the GPU must not be replaced or forked (the reference's counterpart is
    `parallel -j 20` over OS processes, /root/reference/README.md:110-115)."""
    port = a.master_port or _free_port()
    args = [x for x in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + args
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=sys.stderr, cwd=ROOT)
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    if r.returncode == 0 and not lines:
        sys.stderr.write("bench.py: the ranks exited cleanly but rank 0 printed no JSON line\n")
        return 1
    return r.returncode


def chain_lengths(w, idx=None):
    """{parameter vector: largest split index} of the candidates (one chain per distinct parameter vector)."""
    s_int = np.floor(w.split_time).astype(int)
    sel = range(w.n_cand) if idx is None else idx
    if w.params is None:
        return {(): int(max(s_int[i] for i in sel))} if len(sel) else {}
    chains = {}
    for i in sel:
        row = tuple(w.params[i])
        chains[row] = max(chains.get(row, 0), int(s_int[i]))
    return chains


def follow_limit():
    """Chains up to which a batch runs one chain per wave with the trunk wave following (misti_consts.h)."""
    try:
        txt = open(os.path.join(ROOT, "misti_amd", "csrc", "misti_consts.h")).read()
        import re
        m = re.search(r"FOLLOW_MAX_CHAINS\s*=\s*(\d+)", txt)
        return int(m.group(1)) if m else 256
    except Exception:
        return 256


def smoothing_runs(lh, k):
    """(run_start[t], run_end[t]) of genome k: greedy runs of numerically constant lh, as SmoothConst scans them
    (/root/reference/MigrationInference.py:387-405; misti_api.cpp: smoothing_runs)."""
    numT = len(lh)
    rs, re, i = [0] * numT, [0] * numT, 0
    while i < numT:
        j = i
        while j < numT - 1 and abs(lh[j][k] - lh[i][k]) < 1e-10:
            j += 1
        if j == i:
            j = i + 1
        for t in range(i, j):
            rs[t], re[t] = i, j
        i = j
    return rs, re


def trunk_leave_index(w, runs, st):
    """First interval whose rates depend on this candidate's split (misti_kernels.hip: trunk_leave): where it reads its chain's trunk."""
    s = int(np.floor(st))
    frac = st != s
    t_own = s
    if w.flags.get("smooth"):
        if frac:
            t_own = min(runs[0][0][s], runs[1][0][s])
        elif s > 0:
            for k in (0, 1):
                if runs[k][1][s - 1] > s:
                    t_own = min(t_own, runs[k][0][s - 1])
    return t_own


def algorithmic_bytes(w, n_rep, idx=None):
    """HBM bytes the algorithm needs per launch, per kernel.

    correction: one chain per distinct parameter vector, computed up to the largest split index of
                its members: 8P in; per interval 16 B of rates + 48 B of pair state out (+ the trunk records,
                1 056 B per chain and distinct interval at which a member leaves the trunk, when the trunk wave
                follows its chain in the same launch);
    spectrum:   per candidate split 8 + params 8P in, its share of the chain 16 B per two-population
                interval + 48 B state, JAFS 56 + status 4 out (+ one trunk record of 1 056 B when chains are
                shared; the trunk launch itself writes one record per chain and interval);
    llk:        JAFS 56 + status 4 in (the replicate table is shared), 8 per replicate out."""
    P = w.n_param
    sel = np.arange(w.n_cand) if idx is None else np.asarray(idx)
    n = len(sel)
    s_int = np.floor(w.split_time[sel]).astype(int)
    chains = chain_lengths(w, sel)
    correct = sum(8 * P + 64 * L for L in chains.values())
    spectrum = int((8 + 8 * P + 16 * s_int + 48 + 60).sum())
    if len(chains) * 8 <= n:                                  # TRUNK_MIN_SHARE (misti_consts.h): chains are shared, a trunk is built
        # the trunk of a chain hands over one record (state vector + two occupation integrals, 1 056 B) per DISTINCT interval at which
        # some member leaves it
        runs = [smoothing_runs(w.lh, k) for k in (0, 1)]
        key = (lambda i: tuple(w.params[i])) if w.params is not None else (lambda i: ())
        trunk = 1056 * len({(key(i), trunk_leave_index(w, runs, float(w.split_time[i]))) for i in sel})
        spectrum += 1056 * n                                  # one trunk record per candidate
        if len(chains) <= follow_limit():                     # one chain per wave: the trunk wave follows its chain inside the chain launch
            correct += trunk
        else:
            spectrum += trunk
    return {"correct": correct, "spectrum": spectrum, "llk": n * (60 + 8 * n_rep), "n_chains": len(chains)}


def cpu_baseline_child(w, idx, cores, idx_compiled):
    """The CPU baselines in a freshly started child process (the NumPy oracle forks its worker pool there: no fork after
    this process has initialised HIP / torch / RCCL - ADVICE r1; the compiled baseline runs its OpenMP threads there too).
    Returns the child's result arrays: llk, status, runaway, wall of the NumPy oracle on `idx`; c_* of the compiled
    baseline (oracle/cpu/misti_cpu.cpp) on `idx_compiled`."""
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, "in.npz"), os.path.join(tmp, "out.npz")
        np.savez(src, times=np.asarray(w.times, dtype=float), lh=np.asarray(w.lh, dtype=float), split=w.split_time,
                 params=w.params if w.params is not None else np.zeros((0, 0)), jsfs=w.jsfs, idx=np.asarray(idx),
                 idx_compiled=np.asarray(idx_compiled),
                 meta=json.dumps({"bands": w.bands, "pulses": w.pulses, "flags": w.flags, "sample_date": w.sample_date,
                                  "n_param": w.n_param, "cores": cores}))
        env = dict(os.environ)
        env.pop("GPU_MAX_HW_QUEUES", None)
        r = subprocess.run([sys.executable, "-m", "oracle.baseline_child", src, dst], cwd=ROOT, env=env, stdout=sys.stderr, stderr=sys.stderr)
        if r.returncode != 0:
            raise RuntimeError("cpu_baseline child failed (%d)" % r.returncode)
        d = np.load(dst)
        return {k: d[k] for k in d.files}


def dry_run(a, json_fd):
    """Launcher rehearsal on CPU: gloo ranks shard ONE stub grid with misti_amd.dist.evaluate_sharded (the library's
    multi-rank entry point) and gather it.  The stub evaluator is a closed-form function of (split, rate) - neither the
    engine nor the oracle - so this checks process start, rendezvous, sharding, the gather and the one-line protocol."""
    import torch.distributed as dist
    from misti_amd.dist import env_rank, evaluate_sharded, shard_indices
    rank, local_rank, world = env_rank()
    if a.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    if world > 1:
        dist.init_process_group("gloo")
    n = 37                                                     # ragged on purpose
    split = 20.0 + np.arange(n)
    params = (0.01 * (1 + np.arange(n)))[:, None]
    jsfs = np.ones((2, 8))
    stub = lambda s, p, j: np.stack([-(s + p[:, 0]), -(2 * s + p[:, 0])], axis=1)
    t0 = time.perf_counter()
    for _ in range(max(1, a.steps)):
        got = evaluate_sharded(stub, split, params, jsfs).numpy()
    dt = time.perf_counter() - t0
    ok = bool(np.array_equal(got, stub(split, params, jsfs)))
    mine = shard_indices(n, rank, world)
    if rank == 0:
        out = {"metric": "dry-run (launcher rehearsal, no measurement)", "value": None, "unit": "llk evals/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": 1e3 * dt / max(1, a.steps), "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
               "dtype": "f64", "data": "dry-run",
               "config": {"workload": "stub grid of %d candidates x 2 replicates" % n, "world_size": dist.get_world_size() if world > 1 else 1,
                          "backend": "gloo", "candidates_rank0": int(len(mine))},
               "gather_equals_unsharded": ok}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


FIT_OVERRIDE = [""]


EXTRA_LEGS_TIMEOUT_S = 300     # multi-rank runs: wall-clock bound on the secondary legs together (see main())


def build_workload(name, spec):
    from misti_amd import workloads
    w = workloads.BUILDERS[name](spec)
    if FIT_OVERRIDE[0]:
        w.flags = dict(w.flags, cpfit=FIT_OVERRIDE[0] == "cpfit")
        w.name += " [fit overridden: %s]" % FIT_OVERRIDE[0]
    return w


def main():
    a = parse()
    FIT_OVERRIDE[0] = a.fit
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a)
    # stdout carries exactly one JSON line: libraries that print banners there (RCCL prints its version
    # block on communicator creation) are sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if a.dry_run:
        return dry_run(a, json_fd)
    import torch
    import torch.distributed as dist
    from misti_amd.dist import chain_shards, env_rank, shard_indices
    from misti_amd.engine import Engine, Lanes, truth_spectrum

    rank, local_rank, world = env_rank()
    if a.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or a.force_dist          # --force-dist: rehearse the RCCL path with one rank
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
        else:
            dist.init_process_group("nccl", device_id=dev)
    group_world = dist.get_world_size() if use_dist else 1      # what the process group itself reports

    spec = lambda *args: truth_spectrum(*args, device=local_rank)
    if a.workload in ("config3-search", "config3-basinhopping"):
        from misti_amd import search_bench
        return search_bench.run(a, spec, dev, local_rank, rank, world, json_fd)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def leg(workload, scaling, steps, warmup, min_seconds, n_streams, serial_pass=True, keep=False):
        """One measured leg: `steps` steps of `workload` round-robin on n_streams lanes (weak: every rank its own grid;
        strong: ONE grid sharded over the ranks), W warmup steps first, the K-step region repeated until min_seconds."""
        w = build_workload(workload, spec)
        strong = scaling == "strong"
        shards = None
        if strong:
            # ONE grid: whole chains per rank (dist.chain_shards), or interleaved candidates (every chain on every rank)
            sw = a.pretend_world if (a.pretend_world > 1 and world == 1) else world
            shards = chain_shards(w.params, w.n_cand, sw, w.split_time) if a.shard == "chain" else [shard_indices(w.n_cand, r, sw, interleave=True) for r in range(sw)]
            mine = shards[rank]
        else:
            mine = np.arange(w.n_cand)
            if world > 1 and w.params is not None:
                # distinct grids per rank (weak scaling): shift the rate axis by a rank-dependent factor
                w.params = w.params * (1.0 + 0.01 * rank)
        n_total = w.n_cand
        n, R, P = len(mine), int(w.jsfs.shape[0]), w.n_param
        per = max(len(x) for x in shards) if strong else n                      # rows every rank contributes to the gather (ragged shards padded)
        d_split = torch.as_tensor(w.split_time[mine], dtype=torch.float64, device=dev)
        d_par = torch.as_tensor(w.params[mine], dtype=torch.float64, device=dev).contiguous() if P else None
        d_jsfs = torch.as_tensor(w.jsfs, dtype=torch.float64, device=dev).contiguous()
        bucket = max(1, a.gather_bucket) if use_dist else 1
        torch.cuda.synchronize()                          # inputs have landed before any lane (non-blocking streams) reads them

        class Lane:
            """One engine context + its output buffers on one HIP stream.

            A lane issues on its engine's OWN stream (hipStreamCreate inside misti_create): streams created
            one by one map to distinct hardware queues, whereas streams handed out by torch's pool share
            them (measured: ~11 long kernels in flight on 12 own streams against ~6 on 16 pool streams)."""
            def __init__(self, pool, index):
                # lane `index` of the library's own pool (misti_create_lanes, include/misti_hip.h: what a C caller uses to reach this rate);
                # its context is borrowed for the per-lane stream handle and the per-kernel timing of the serial pass
                self.pool, self.index = pool, index
                self.eng = pool.engine(index)
                self.stream = torch.cuda.ExternalStream(self.eng.stream_handle(), device=dev)
                # multi-GPU: the llk of `bucket` consecutive batches of this lane are gathered by ONE collective (a 32 KB all_gather
                # per batch is pure latency on xGMI and costs a fifth of the rate); every batch's llk still reaches every rank
                self.slots = torch.full((bucket, per, R), float("nan"), dtype=torch.float64, device=dev)   # rows beyond n: padding of a ragged shard
                self.gathered = torch.empty((world * bucket * per, R), dtype=torch.float64, device=dev) if use_dist else None
                self.fill = 0
                self.llk = self.slots[0]
                self.jafs = torch.empty((n, 7), dtype=torch.float64, device=dev)
                self.status = torch.empty(n, dtype=torch.int32, device=dev)
                # one batch per slot of the bucket, arguments converted once (Lanes.bind_dev: the step is the same library call, 2 us shorter)
                self.issue = [pool.bind_dev(index, n, d_split.data_ptr(), d_par.data_ptr() if P else 0, R, d_jsfs.data_ptr(),
                                            self.slots[k].data_ptr(), self.jafs.data_ptr(), 0, 0, self.status.data_ptr()) for k in range(bucket)]

            def step(self):
                self.llk = self.slots[self.fill]
                self.issue[self.fill]()
                if use_dist:
                    self.fill += 1
                    if self.fill == bucket:
                        self.flush()

            def flush(self, filled_only=False):
                """The bucket's collective, issued from the lane's own stream: c10d orders it after the batches (event on this
                stream), runs it on its communicator stream in host issue order - the same on every rank - and makes this
                stream wait for it, so the next batch of the lane cannot overwrite a slot early.  (A dedicated communication
                stream + events per step cost a hardware queue and 15 % of the rate.)  A partly filled bucket is gathered whole
                unless `filled_only` (the end of a region: flush_all)."""
                if use_dist and self.fill > 0:
                    rows = (self.fill if filled_only else bucket) * per
                    with torch.cuda.stream(self.stream):
                        dist.all_gather_into_tensor(self.gathered[: world * rows], self.slots.view(bucket * per, R)[:rows])
                    self.fill = 0

            def close(self):
                self.eng.sync()
                self.stream = None

        host_issue = [0.0]

        def flush_all(lanes):
            """What is still in the lanes' buckets at the end of a region: every lane gathers the FILLED slots of its bucket with a
            collective of its own, lanes with fewer batches first.  (Until round 6: the whole bucket of 8 whatever it held - in the
            driver's shape, 20 steps on 18 lanes, one or two batches: 8 x and 4 x the bytes over xGMI - and in lane order, which put all
            eighteen collectives behind lane 0, the lane that is two batches deep: c10d runs them in host order on ONE communicator
            stream.  `--flush-mode each` is that form.)  ONE grouped collective for all lanes (c10d's coalescing manager, issued from one
            lane's stream after it waited for the others) was built and measured on one rank with RCCL initialised: 1.37e7 against
            2.25e7 evals/s in the driver's shape, and 1.06e7 with the same collectives ungrouped - eighteen cross-queue waits cost
            more than eighteen small collectives (profiles/r06_flush_modes.txt); not kept.  Nor is the form whose lanes do not wait for
            their collective (async_op, released behind the fence): + 3.6 % with 24 hardware queues, - 8 % under the cap of 22 (it keeps
            one more queue busy)."""
            todo = [lane for lane in lanes if lane.fill > 0]
            if not use_dist or not todo:
                return
            if a.flush_mode == "each":
                for lane in todo:
                    lane.flush()
                return
            for lane in sorted(todo, key=lambda l: l.fill):           # stable sort: the same order on every rank
                lane.flush(filled_only=True)

        def timed(lanes, k):
            """EXACTLY `k` steps issued round-robin on `lanes` between two fences; seconds (max over ranks)."""
            fence()
            t0 = time.perf_counter()
            for i in range(k):
                lanes[i % len(lanes)].step()
            flush_all(lanes)                             # what is still in a bucket belongs to the timed steps
            host_issue[0] = time.perf_counter() - t0
            fence()
            dt = time.perf_counter() - t0
            if use_dist:
                t = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            return dt

        def repeated(lanes, k, wu, min_s):
            """W untimed warmup steps, then the K-step timed region again and again until the regions add up to
            min_seconds (at least once, at most 200 times; every rank takes the same decision: dt is the all-reduced max)."""
            for i in range(wu):
                lanes[i % len(lanes)].step()
            flush_all(lanes)
            dts = []
            while not dts or (sum(dts) < min_s and len(dts) < 200):
                dts.append(timed(lanes, k))
            return dts

        # what a caller sweeping a grid knows about its own batches: the split times of this workload have no fractional part (the device
        # verifies it per candidate); config 4's scan has fractional splits and says nothing
        pool = Lanes(w.times, w.lh, device=local_rank, lanes=max(1, n_streams), **w.engine_kwargs())
        pool.set_hints(integer_splits=bool(np.all(w.split_time == np.floor(w.split_time))))
        lanes = [Lane(pool, i) for i in range(pool.n_lanes)]
        # context initialisation, not measurement: the first batch of a context allocates its workspaces (hipMalloc) and
        # learns its launch shape; every lane does that once here so that a short run (K < lanes x a few) times steady state
        for lane in lanes:
            lane.step()
        fence()
        dts = repeated(lanes, steps, warmup, min_seconds)
        dt = statistics.median(dts)
        res = {"w": w, "mine": mine, "n": n, "n_total": n_total, "R": R, "P": P, "strong": strong, "bucket": bucket,
               "dts": dts, "dt": dt, "issue": host_issue[0], "steps": steps, "n_streams": len(lanes)}
        job_cands = n_total if strong else world * n          # candidates all ranks evaluate per step
        if strong and a.pretend_world > 1 and world == 1:
            job_cands = n                                     # a per-rank cost model: only rank 0's shard is evaluated, only it is counted
        res["job_cands"] = job_cands
        res["value"] = job_cands * R * steps / dt
        if serial_pass:
            # strictly serial pass on one stream: the per-grid latency, and the per-kernel durations (HIP events on the launch stream)
            serial = lanes[0]
            eng = serial.eng
            k_serial = max(4, min(steps, 16))
            dts_serial = repeated([serial], k_serial, 2, min_seconds)
            res["dt_serial"], res["k_serial"], res["repeats_serial"] = statistics.median(dts_serial), k_serial, len(dts_serial)
            eng.enable_timing(True)
            eng.kernel_times(reset=True)
            timed([serial], k_serial)
            res["kms"], res["kn"] = eng.kernel_times(reset=True)
            eng.enable_timing(False)
        res["status"] = lanes[0].status.cpu().numpy()
        res["llk"] = lanes[0].llk[:n].cpu().numpy()
        chains_mine = len(chain_lengths(w, mine))
        if use_dist:
            t = torch.zeros(world, dtype=torch.int64, device=dev)
            t[rank] = chains_mine
            dist.all_reduce(t)
            res["chains_per_rank"] = [int(v) for v in t.cpu()]
            cand_t = torch.zeros(world, dtype=torch.int64, device=dev)
            cand_t[rank] = n
            dist.all_reduce(cand_t)
            res["cands_per_rank"] = [int(v) for v in cand_t.cpu()]
        else:
            res["chains_per_rank"], res["cands_per_rank"] = [chains_mine], [n]
        if keep:
            res["lanes"], res["pool"] = lanes, pool
        else:
            for lane in lanes:
                lane.close()
            pool.close()
        return res

    def stored_llk_traffic():
        """WRITE_SIZE + 2 x FETCH_SIZE of the large launch from the stored --pmc passes, if they belong to THIS build."""
        try:
            j = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            return j["llk"]["hbm_bytes_per_launch"] if j.get("build_id") == _build_id() and "llk" in j else None
        except Exception:                      # noqa: BLE001
            return None

    def llk_roofline_leg(n_cand=65536, n_rep=1000, reps=20):
        """The replicate epilogue alone (the one HBM-write-bound kernel of the path, MigrationInference.py:600-609) at a size
        where the roofline means something: `misti_llk_dev` on n_cand spectra x n_rep bootstrap replicates - 8 bytes written per value
        (524 MB at the default), 56 in per candidate, 72 in per replicate.  Duration: HIP events on the launch stream around the
        kernel (misti_kernel_times[2]).  Also at BASELINE config 4's own size (256 x 1 000), where the launch dominates."""
        rng = np.random.default_rng(11)
        out_legs = {}
        w4 = build_workload("config4", spec)
        with Engine(w4.times, w4.lh, device=local_rank, **w4.engine_kwargs()) as eng:
            rows = np.ascontiguousarray(np.resize(w4.jsfs, (n_rep, 8)), dtype=np.float64)
            d_rows = torch.as_tensor(rows, device=dev)
            for tag, nc in (("config4", 256), ("large", n_cand)):          # the large one last: a rocprofv3 --pmc pass keeps a kernel's LAST launch
                jf = rng.random((nc, 7)) + 0.05
                jf /= jf.sum(axis=1, keepdims=True)
                d_jafs = torch.as_tensor(jf, device=dev)
                d_llk = torch.empty((nc, n_rep), dtype=torch.float64, device=dev)
                torch.cuda.synchronize()
                for _ in range(3):
                    eng.llk_dev(nc, d_jafs.data_ptr(), 0, n_rep, d_rows.data_ptr(), d_llk.data_ptr())
                eng.sync()
                eng.enable_timing(True)
                eng.kernel_times(reset=True)
                for _ in range(reps):
                    eng.llk_dev(nc, d_jafs.data_ptr(), 0, n_rep, d_rows.data_ptr(), d_llk.data_ptr())
                eng.sync()
                kms, kn = eng.kernel_times(reset=True)
                eng.enable_timing(False)
                ms = kms["llk"] / max(1, kn["llk"])
                nbytes = 8.0 * nc * n_rep + 56.0 * nc + 72.0 * n_rep
                # the values themselves, against NumPy on a sample of rows (the parity tests hold the kernel to the bits of the inline epilogue)
                got = d_llk[:: max(1, nc // 16)].cpu().numpy()
                lj = np.log(np.stack([jf[:, 0] + jf[:, 6], jf[:, 1] + jf[:, 5], jf[:, 2] + jf[:, 4], jf[:, 3]], axis=1))[:: max(1, nc // 16)]
                ff = np.stack([rows[:, 1] + rows[:, 7], rows[:, 2] + rows[:, 6], rows[:, 3] + rows[:, 5], rows[:, 4]], axis=1)
                from scipy.special import gammaln
                cst = gammaln(rows[:, 1:].sum(axis=1) + 1) - gammaln(ff + 1).sum(axis=1)
                want = cst[None, :] + lj @ ff.T
                out_legs[tag] = {"kernel": "misti::llk_kernel", "spectra": nc, "replicates": n_rep, "ms_per_launch": ms, "launches": int(kn["llk"]),
                                 "algorithmic_bytes_per_launch": nbytes, "bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_of_achievable_6300": nbytes / (ms * 1e-3) / 1e9 / 6300.0,
                                 "llk_per_s": nc * n_rep / (ms * 1e-3), "max_rel_err_vs_numpy": float(np.max(np.abs(got / want - 1.0))),
                                 "traffic": stored_llk_traffic() if tag == "large" else None}
                del d_llk, d_jafs
        out_legs["note"] = ("the replicate epilogue alone (misti_llk_dev): algorithmic bytes = 8 per value + 56 per spectrum + 72 per replicate; duration = HIP events "
                            "around the kernel on its stream; peak 8 TB/s (6.3 TB/s is what a plain float4 copy reaches on this chip: MI355X_MICROARCH.md); "
                            "`traffic`: WRITE_SIZE / FETCH_SIZE of the same launch, profiles/rNN_pmc_llk.json")
        return out_legs

    def host_abi_leg(workload, k):
        """The C ABI's HOST-buffer form, misti_eval_batch (pageable NumPy arrays in and out, as a ctypes caller of the reference
        would hold them; PCIe both ways inside the call, which returns when the results are in the caller's memory): K calls
        one after another on one context, then K calls from two host threads on two contexts (the copies of one batch
        overlap the kernels of the other; ctypes releases the GIL for the duration of a call)."""
        import ctypes as C
        import threading
        from misti_amd import _lib as L
        w = build_workload(workload, spec)
        n, R, P = w.n_cand, int(w.jsfs.shape[0]), w.n_param
        lib = L.load()
        ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None

        class Ctx:
            def __init__(self):
                self.eng = Engine(w.times, w.lh, device=local_rank, **w.engine_kwargs())
                self.split = np.ascontiguousarray(w.split_time, dtype=np.float64)
                self.par = np.ascontiguousarray(w.params, dtype=np.float64) if P else None
                self.rows = np.ascontiguousarray(w.jsfs, dtype=np.float64)
                self.llk, self.jafs, self.status = np.empty((n, R)), np.empty((n, 7)), np.empty(n, dtype=np.int32)

            def call(self):
                L.check(lib.misti_eval_batch(self.eng._ctx, n, ptr(self.split), ptr(self.par), None, R, ptr(self.rows), ptr(self.llk), ptr(self.jafs),
                                             None, None, ptr(self.status)))

        a_, b_ = Ctx(), Ctx()
        for c in (a_, b_):
            for _ in range(3):
                c.call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            a_.call()
        serial = (time.perf_counter() - t0) / k

        def worker(c, m):
            for _ in range(m):
                c.call()
        th = [threading.Thread(target=worker, args=(c, k // 2)) for c in (a_, b_)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        both = (time.perf_counter() - t0) / (2 * (k // 2))
        ok = float((a_.status == 0).mean())
        in_b = 8 * n * (1 + P) + 64 * R
        out_b = 8 * n * R + 56 * n + 4 * n
        a_.eng.close(); b_.eng.close()
        return {"workload": w.name, "entry_point": "misti_eval_batch (host buffers, pageable; pinned staging inside the library)",
                "ms_per_batch": 1e3 * serial, "value": n * R / serial, "unit": "llk evals/s", "calls": k, "contexts": 1,
                "two_contexts": {"ms_per_batch": 1e3 * both, "value": n * R / both, "contexts": 2, "host_threads": 2,
                                 "note": "two host threads, one context each: the copies of one batch overlap the kernels of the other"},
                "bytes_in": in_b, "bytes_out": out_b, "status_ok_fraction": ok,
                "note": "PCIe-inclusive: never `value` of the line; compare with single_batch (device-resident inputs, same grid)"}

    def block(res, note=None):
        """A secondary leg as a JSON block."""
        b = {"workload": res["w"].name, "value": res["value"], "unit": "llk evals/s", "ms_per_step": 1e3 * res["dt"] / res["steps"], "steps": res["steps"],
             "streams": res["n_streams"], "candidates_total": res["job_cands"], "replicates": res["R"], "repeats": len(res["dts"]),
             "candidates_per_rank": res["cands_per_rank"], "chains_per_rank": res["chains_per_rank"],
             "status_ok_fraction": float((res["status"] == 0).mean())}
        if "dt_serial" in res:
            b["single_batch_ms"] = 1e3 * res["dt_serial"] / res["k_serial"]
            b["ms_per_launch"] = {k: (res["kms"][k] / res["kn"][k] if res["kn"][k] else 0.0) for k in res["kms"]}
        if note:
            b["note"] = note
        return b

    n_streams = max(1, a.streams)
    main_leg = leg(a.workload, a.scaling, a.steps, a.warmup, a.min_seconds, n_streams)
    w, mine, n, R, dt, dts = main_leg["w"], main_leg["mine"], main_leg["n"], main_leg["R"], main_leg["dt"], main_leg["dts"]
    strong, bucket, job_cands, value = main_leg["strong"], main_leg["bucket"], main_leg["job_cands"], main_leg["value"]
    status, llk = main_leg["status"], main_leg["llk"]
    kms, kn, dt_serial, k_serial = main_leg["kms"], main_leg["kn"], main_leg["dt_serial"], main_leg["k_serial"]

    metric = "composite-llk evals/sec over (split×mi) grid, 128 merged PSMC intervals"
    try:                                              # the exact string of BASELINE.json when it is at hand
        metric = json.load(open(os.path.join(ROOT, "BASELINE.json"))).get("metric") or metric
    except Exception:
        pass
    out = {
        "metric": metric,
        "value": value, "unit": "llk evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "library": {"build_id": _build_id(), "abi": _abi()},
        "config": {"workload": w.name, "candidates_per_gpu": n, "candidates_total": job_cands, "replicates": R, "numT": w.numT,
                   "batches_in_flight": n_streams, "streams": n_streams,
                   "world_size": group_world, "candidates_per_rank": main_leg["cands_per_rank"], "chains_per_rank": main_leg["chains_per_rank"],
                   "gather_bucket": bucket if use_dist else None, "pretend_world": a.pretend_world if (a.pretend_world > 1 and world == 1 and strong) else None,
                   "parallelism": ("ONE grid sharded over the ranks (%s), all_gather of llk (RCCL), one collective per %d batches of a lane" % ("whole chains per rank" if a.shard == "chain" else "interleaved", bucket) if strong else
                                   "every rank its own grid, all_gather of llk (RCCL), one collective per %d batches of a lane" % bucket) if world > 1 else "1 GPU"},
        "timing": {"repeats": len(dts), "timed_region_s_median": dt, "timed_region_s_total": sum(dts),
                   "timed_region_s_min": min(dts), "timed_region_s_max": max(dts),
                   "note": "value and ms_per_step are the MEDIAN repetition of the K-step timed region (each between barriers, max over ranks)"},
    }
    out_spectra_per_s = job_cands * a.steps / dt

    # ---- secondary legs (every rank runs them: they contain collectives) -------------------------------------------------
    # A secondary leg must never cost the headline line.  With several ranks a leg that raises on ONE rank only (an out-of-memory on one
    # GPU, a HIP error) leaves the others inside a collective for ever: a watchdog on every rank bounds the legs - when it
    # expires rank 0 emits the headline line as it stands (the legs marked as timed out) and every rank leaves with exit status 3: the
    # measured headline is on stdout, and a launcher can still tell a run that hung in a secondary leg from a clean one.
    extra = {}
    watchdog = None
    if not a.no_extra_legs and world > 1:
        import threading
        headline_only = dict(out, extra_legs={"error": "a secondary leg did not finish within %d s on every rank; headline leg only" % EXTRA_LEGS_TIMEOUT_S})

        def expire():
            if rank == 0:
                os.write(json_fd, (json.dumps(headline_only) + "\n").encode())
            os._exit(3)
        watchdog = threading.Timer(EXTRA_LEGS_TIMEOUT_S, expire)
        watchdog.daemon = True
        watchdog.start()
    if not a.no_extra_legs:
        short = max(8, min(a.steps, 64))
        if world > 1:
            # both scalings in one line: the headline leg is one of them, the other one here
            other = "strong" if a.scaling == "weak" else "weak"
            extra[a.scaling] = block(main_leg, "the headline leg above")

            def any_rank_failed(failed):
                # A rank that caught an exception in one leg must not walk into the NEXT leg's collectives while its peers are still
                # inside this one's (collectives of different sizes would be paired: ADVICE r4): after every leg the ranks agree on
                # a failure flag, and one failure anywhere skips the remaining collective legs everywhere.  A rank stuck INSIDE a
                # collective never gets here - that is what the watchdog above is for; it ends every rank with exit status 0 because
                # the headline line is already valid, and says so in `extra_legs.error` (the driver reads the line, not the status).
                flag = torch.tensor([1.0 if failed else 0.0], dtype=torch.float64, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                return bool(flag.item() > 0)
            leg_failed = False
            try:                                   # a secondary leg must never cost the headline line (the same code runs on every rank)
                o_leg = leg("config5" if other == "strong" else "config2", other, short, min(a.warmup, 16), a.min_seconds, n_streams, serial_pass=False)
                extra[other] = block(o_leg, "ONE config-5 grid (65 536 candidates, 2 048 chains) sharded over the ranks, %s" % ("whole chains per rank" if a.shard == "chain" else "interleaved") if other == "strong"
                                     else "every rank its own config-2 grid")
            except Exception as e:                 # noqa: BLE001
                extra[other] = {"error": "%s: %s" % (type(e).__name__, e)}
                leg_failed = True
            skip_rest = any_rank_failed(leg_failed)
            # ... and BASELINE config 3 as a search, strong: ONE Nelder-Mead search of 16 384 starts, the starts dealt to the ranks in
            # contiguous blocks, one all_gather of the results (misti_amd.optimize.solve_batched_dev inside the process group)
            try:
                if skip_rest:
                    raise RuntimeError("skipped: an earlier secondary leg failed on some rank")
                from misti_amd.optimize import solve_batched_dev
                w3 = build_workload("config3", spec)
                with Engine(w3.times, w3.lh, device=local_rank, **w3.engine_kwargs()) as eng3:
                    search = lambda: solve_batched_dev(eng3, float(w3.split_time[0]), w3.params, w3.jsfs[0], tol=1e-4, maxiter=1000)[2]
                    search()                       # allocates the search state
                    fence()
                    t0 = time.perf_counter()
                    r3 = search()
                    fence()
                    dt3 = time.perf_counter() - t0
                t3 = torch.tensor([dt3], dtype=torch.float64, device=dev)
                dist.all_reduce(t3, op=dist.ReduceOp.MAX)
                dt3 = float(t3.item())
                extra["strong_search"] = {"workload": "config3-search: Nelder-Mead from %d random starts, starts dealt to the ranks in contiguous blocks, one all_gather" % w3.n_cand,
                                          "value": float(r3["nfev"].sum()) / dt3, "unit": "objective evaluations/s incl. optimiser", "s_per_search": dt3,
                                          "starts": int(w3.n_cand), "starts_per_rank": -(-int(w3.n_cand) // world), "converged_fraction": float((r3["status"] == 0).mean())}
            except Exception as e:                 # noqa: BLE001
                extra["strong_search"] = {"error": "%s: %s" % (type(e).__name__, e)}
        elif a.workload == "config2":
            # ONE call on ONE stream at the overlapped rate: 16 config-2 grids with distinct rate axes in one batch
            try:
                s_leg = leg("config2x16", "weak", max(4, min(a.steps, 32)), 4, a.min_seconds, 1, serial_pass=True)
                extra["single_call"] = block(s_leg, "16 config-2 grids with distinct rate axes (65 536 candidates, 1 024 chains) as ONE misti_eval_batch_dev "
                                                    "call per step, strictly one after another on ONE stream")
            except Exception as e:                 # noqa: BLE001
                extra["single_call"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                extra["host_abi"] = host_abi_leg("config2", max(16, min(a.steps, 128)))
            except Exception as e:                 # noqa: BLE001
                extra["host_abi"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and a.workload in ("config2", "config4"):
            try:
                extra["llk_kernel"] = llk_roofline_leg()
            except Exception as e:                 # noqa: BLE001
                extra["llk_kernel"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if watchdog is not None:
        fence()                                    # every rank is through its secondary legs
        watchdog.cancel()
    if rank == 0:
        out.update(extra)
        out["single_batch"] = {"value": job_cands * R * k_serial / dt_serial, "ms_per_step": 1e3 * dt_serial / k_serial, "steps": k_serial,
                               "streams": 1, "repeats": main_leg["repeats_serial"],
                               "note": "the same step issued strictly one after another on one stream: one grid's latency, bounded by its "
                                       "longest lambda-correction chain (up to ~900 dependent residual evaluations in the runaway corner of "
                                       "the grid); `value` above overlaps config.batches_in_flight such batches"}
        out["serial"] = out["single_batch"]          # round-1 name of the same block
        # ---- roofline of the dominant kernel -----------------------------------------
        per_ms = {k: (kms[k] / kn[k] if kn[k] else 0.0) for k in kms}          # ms per launch, HIP events on the launch stream
        dom = max(("correct", "spectrum"), key=lambda k: per_ms[k])
        ab = algorithmic_bytes(w, R, mine)
        achieved = ab[dom] / (per_ms[dom] * 1e-3) / 1e9 if per_ms[dom] > 0 else 0.0
        out["roofline"] = roofline_block(a.workload, world, dom, per_ms, ab, achieved, n)
        if "llk_kernel" in out:
            out["roofline"]["llk_kernel"] = out.pop("llk_kernel")       # the HBM-write-bound kernel reported separately
        v = out["roofline"].get("valu")
        if v and v.get("wave_insts"):
            # the same counters at the rate of the headline leg: launches of the dominant kernel per second x its wave-instructions x 4 cycles,
            # against all SIMD cycles of the chip - what share of the fp64 issue slots the OVERLAPPED run keeps busy
            launches_per_s = a.steps / dt
            second_insts = (out["roofline"].get("second_kernel") or {}).get("wave_insts") or 0.0
            v["frac_at_value_dominant_kernel"] = 4.0 * v["wave_insts"] * launches_per_s / (N_SIMD * GPU_CLOCK_HZ)
            v["frac_at_value"] = 4.0 * (v["wave_insts"] + second_insts) * launches_per_s / (N_SIMD * GPU_CLOCK_HZ)
            v["frac_at_value_note"] = ("chip-wide share of fp64 VALU issue slots during the timed region of `value` (%d batches in flight): wave-instructions of "
                                       "the dominant kernel AND of the batch's other large kernel, x 4 cycles, x launches per second" % n_streams)
        # secondary figure SURVEY 8d asks for: the flop model of an UNSHARED evaluation (what the reference computes per
        # candidate: ~16 sparse generator applications of 2 x 220 flop per two-population interval, ~5 kflop of 3x3
        # exponentials per migrating interval, 0.1 kflop per one-population interval) x distinct spectra per second,
        # against the fp64 vector peak.  Chain and trunk sharing EXECUTE far fewer flops than this model counts.
        s_int = np.floor(w.split_time).astype(np.int64)
        flop_per_spectrum = float(np.mean(s_int * (16 * 440 + 5000) + (w.numT - s_int) * 100))
        eq_tflops = out_spectra_per_s * flop_per_spectrum / 1e12
        out["roofline"]["fp64_equivalent"] = {"flop_per_spectrum_model": flop_per_spectrum, "equivalent": eq_tflops, "peak": FP64_VALU_PEAK_TFLOPS,
                                              "unit": "TFLOP/s", "frac": eq_tflops / FP64_VALU_PEAK_TFLOPS,
                                              "note": "reference-equivalent work per second (SURVEY 8d model of an unshared evaluation), not executed flops; no credit claimed"}
        ok = status == 0
        out["status_fraction"] = {"ok": float(ok.mean()), "correction_failed": float((status == 2).mean()),
                                  "stiff": float((status == 6).mean()), "numeric": float((status == 5).mean())}
        out["spectrum_evals_per_s"] = out_spectra_per_s
        out["host_issue_ms_per_step"] = 1e3 * main_leg["issue"] / a.steps
        # ---- CPU baseline: the oracle on this box's host cores, bounded sample, in a child process ----------
        if world == 1 and not a.no_cpu_baseline:
            cores = min(os.cpu_count() or 1, 16)
            per_core = 12.0                                                  # evals/s/core, order of magnitude (measured ~18)
            m = int(max(cores, min(n, a.cpu_seconds * per_core * cores)))
            idx = np.linspace(0, n - 1, m).astype(np.int64)
            # the compiled baseline is ~50 x faster per core: a larger sample (the whole grid where that fits the budget)
            mc = int(max(cores, min(n, a.cpu_seconds * 90.0 * cores)))
            idx_c = np.linspace(0, n - 1, mc).astype(np.int64)
            res = cpu_baseline_child(w, mine[idx], cores, mine[idx_c])
            o_llk, o_status, o_run, wall = res["llk"], res["status"], res["runaway"], float(res["wall"])
            c_wall = float(res["c_wall"])
            out["cpu_baseline"] = {"value": len(idx_c) * R / c_wall, "unit": "llk evals/s", "cores": cores, "kind": "port",
                                   "sample": "%d of the %d candidates (evenly spaced) x %d replicate(s); compiled restatement of the reference's algorithm "
                                             "(oracle/cpu/misti_cpu.cpp: C++17 + OpenMP, dense Pade expm + LU per interval, SciPy's TRF restated; pinned on "
                                             "the reference's golden vectors), one candidate per OpenMP task, %.1f s wall" % (len(idx_c), n, R, c_wall)}
            out["cpu_baseline_numpy"] = {"value": len(idx) * R / wall, "unit": "llk evals/s", "cores": cores, "kind": "port",
                                         "sample": "%d of the %d candidates (evenly spaced) x %d replicate(s), NumPy/SciPy oracle (the reference's own "
                                                   "SciPy calls), one process per core (forked by a child process that never touched the GPU), %.1f s wall"
                                                   % (len(idx), n, R, wall)}
            cb = (res["c_status"] == 0) & (status[idx_c] == 0)
            crel = np.abs(llk[idx_c, 0] - res["c_llk"][:, 0]) / np.abs(res["c_llk"][:, 0])
            creg = cb & (res["c_runaway"] < 5.0)
            out["parity_vs_compiled_baseline_sample"] = {
                "n": int(cb.sum()), "status_agree": float((res["c_status"] == status[idx_c]).mean()),
                "regular": {"n": int(creg.sum()), "max_rel": float(crel[creg].max()) if creg.any() else None,
                            "frac_within_1e-9": float((crel[creg] <= 1e-9).mean()) if creg.any() else None}}
            # the same sample doubles as an end-to-end parity check of this run
            both = (o_status == 0) & (status[idx] == 0)
            rel = np.abs(llk[idx, 0] - o_llk[:, 0]) / np.abs(o_llk[:, 0])
            regular = both & (o_run < 5.0)                           # the reference itself is determined
            runaway = both & ~regular
            out["parity_vs_oracle_sample"] = {
                "n": int(both.sum()), "status_agree": float((o_status == status[idx]).mean()),
                "regular": {"n": int(regular.sum()), "max_rel": float(rel[regular].max()) if regular.any() else None,
                            "frac_within_1e-9": float((rel[regular] <= 1e-9).mean()) if regular.any() else None},
                "runaway_rate_candidates": {"n": int(runaway.sum()), "max_rel": float(rel[runaway].max()) if runaway.any() else None,
                                            "note": "reference-indeterminate (corrected rate x interval length >= 5): the reference's own value "
                                                    "moves by comparable amounts under a 2^-48 perturbation of its inputs (tests/parity.py, "
                                                    "profiles/r02_*parity*)"}}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()
    return 0


def _build_id():
    from misti_amd import _lib as _mlib
    return _mlib.build_id()


def _abi():
    from misti_amd import _lib as _mlib
    return int(_mlib.load().misti_abi_version())


def roofline_block(workload, world, dom, per_ms, ab, achieved, n):
    """`roofline` of the dominant kernel.

    The contract's bounds are hbm | mfma; neither binds this path (~1 KB and ~1e5 dependent fp64 operations per
    candidate, 3x3 / sparse 44x44 matrices).  What binds is fp64 VALU issue: the top level reports THAT (bound "valu":
    wave-instructions issued x 4 cycles against 1 024 SIMDs x clock), `hbm` carries the figures the contract asks for
    (algorithmic bytes / HIP-event duration against 8 TB/s, counter traffic), `valu` the raw counters.  Durations are HIP events
    of this run; the counters are rocprofv3 --pmc passes of the same command, stored per workload in profiles/pmc_latest.json."""
    traffic, traffic_source, valu, second = None, None, None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc) and world == 1:
        try:
            j = json.load(open(pmc))
            j = j.get("workloads", {}).get(workload) or (j if j.get("workload") == workload else None)
            if j:
                from misti_amd import _lib as _mlib
                stored_id, mine_id = j.get("build_id"), _mlib.build_id()
                if stored_id != mine_id:
                    # counters of ANOTHER build say nothing about the library that ran: no traffic, no VALU fraction
                    traffic_source = "stale: profiles/pmc_latest.json holds counters of build %s, the loaded library is build %s" % (stored_id, mine_id)
                    j = None
            if j:
                traffic = j.get(dom + "_hbm_bytes_per_launch")
                traffic_source = "stored: profiles/pmc_latest.json <- " + str(j.get("source")) + " (not measured in this run)"
                v = j.get(dom + "_valu")
                if v and v.get("SQ_INSTS_VALU") and per_ms[dom] > 0:
                    # 4 SIMDs x 256 CUs; a wave-instruction in fp64 occupies its SIMD's VALU for 4 cycles (16 lanes per cycle)
                    cycles = per_ms[dom] * 1e-3 * GPU_CLOCK_HZ
                    fp64 = None
                    if v.get("SQ_INSTS_VALU_FMA_F64") is not None:
                        # fp64 priced as fp64: wave-instructions by class (separate --pmc pass); an FMA is two flops per live lane, an add, a
                        # multiply or a transcendental (rcp / rsq / ...) one.  Lanes live per instruction from lane_occupancy (all VALU classes).
                        add, mul, fma, tr = (v.get("SQ_INSTS_VALU_%s_F64" % c, 0.0) for c in ("ADD", "MUL", "FMA", "TRANS"))
                        occ = v.get("lane_occupancy") or 1.0
                        flops = (add + mul + tr + 2.0 * fma) * 64.0 * occ
                        tfl = flops / (per_ms[dom] * 1e-3) / 1e12
                        fp64 = {"wave_insts_fp64": add + mul + fma + tr, "add": add, "mul": mul, "fma": fma, "trans": tr,
                                "wave_insts_int": v.get("SQ_INSTS_VALU_INT32", 0.0) + v.get("SQ_INSTS_VALU_INT64", 0.0), "wave_insts_cvt": v.get("SQ_INSTS_VALU_CVT"),
                                "share_of_valu_insts": (add + mul + fma + tr) / v["SQ_INSTS_VALU"],
                                "flops_per_launch": flops, "achieved": tfl, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / FP64_VALU_PEAK_TFLOPS,
                                "issue_frac_fp64_only": 4.0 * (add + mul + fma + tr) / (N_SIMD * cycles),
                                "note": "executed fp64 flops of this launch (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 lanes x lanes live; FMA = 2) against the "
                                        "78.6 TFLOP/s fp64 vector peak; issue_frac_fp64_only prices only those instructions at 4 cycles - the rest of "
                                        "SQ_INSTS_VALU (integer, 32-bit, moves, compares) issues in fewer, so `frac` above is an upper bound on fp64 use"}
                    valu = {"wave_insts": v["SQ_INSTS_VALU"], "lane_occupancy": v.get("lane_occupancy"), "fp64": fp64,
                            "frac": 4.0 * v["SQ_INSTS_VALU"] / (N_SIMD * cycles),
                            "kernel_cycles": cycles, "clock_hz": GPU_CLOCK_HZ, "simds": N_SIMD, "kernel": v.get("kernel"),
                            "waves_launched": v.get("SQ_WAVES"), "busy_cycles": v.get("SQ_BUSY_CYCLES"),
                            "issue_stall_share": (v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]) if v.get("SQ_WAVE_CYCLES") and v.get("SQ_WAIT_INST_ANY") is not None else None,
                            "source": "SQ counters stored in profiles/pmc_latest.json (rocprofv3 --pmc passes of this command), duration measured in this run",
                            "note": "frac = 4 x SQ_INSTS_VALU / (1 024 SIMDs x kernel cycles at 2.4 GHz): share of the chip's fp64 VALU issue slots this launch "
                                    "used; lane_occupancy = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): lanes live per issued instruction (wave-uniform "
                                    "bookkeeping of a chain runs in all 64 lanes: live, not useful)"}
                # the other big kernel of the batch (kernel 2 when the correction dominates): the same two figures
                oth = "spectrum" if dom == "correct" else "correct"
                vo = j.get(oth + "_valu")
                if vo and vo.get("SQ_INSTS_VALU") and per_ms.get(oth, 0) > 0:
                    cyc = per_ms[oth] * 1e-3 * GPU_CLOCK_HZ
                    second = {"kernel": vo.get("kernel"), "ms_per_launch": per_ms[oth], "wave_insts": vo["SQ_INSTS_VALU"],
                              "valu_frac": 4.0 * vo["SQ_INSTS_VALU"] / (N_SIMD * cyc), "lane_occupancy": vo.get("lane_occupancy"),
                              "hbm_traffic_bytes": j.get(oth + "_hbm_bytes_per_launch"),
                              "hbm_frac": (j.get(oth + "_hbm_bytes_per_launch") or 0.0) / (per_ms[oth] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "note": "the batch's other large kernel, same definitions (counter traffic / duration for hbm_frac)"}
        except Exception:
            traffic, valu, second = None, None, None
    hbm = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
           "algorithmic_bytes_per_launch": ab[dom]}
    kernel_name = (valu or {}).get("kernel") or ("correct_follow_kernel (<= 1 024 chains: one chain per wave) / correct_kernel + correct_resume_kernel (packed)"
                                                 if dom == "correct" else "spectrum_kernel")
    common = {"kernel": kernel_name, "candidates_per_launch": n, "chains_per_launch": ab["n_chains"], "ms_per_launch": per_ms, "traffic": traffic,
              "traffic_source": traffic_source, "algorithmic_bytes_per_launch": ab[dom], "hbm": hbm,
              "binding_resource": "fp64 VALU issue: the dependent instruction stream of each lambda-correction chain (serial trust-region iterations of the "
                                  "reference's solver) - latency-bound with few chains per launch, issue-bound with many; not HBM, not MFMA"}
    if valu:
        peak = N_SIMD * GPU_CLOCK_HZ
        blk = {"bound": "valu", "achieved": valu["frac"] * peak, "peak": peak, "unit": "fp64 VALU issue cycles/s (wave-instructions x 4)", "frac": valu["frac"],
               "valu": valu,
               "note": "the contract's two bounds are hbm | mfma and neither binds (SURVEY 8d); `hbm` below carries the HBM figures as asked, the top level "
                       "the resource that binds"}
    else:
        blk = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
               "note": "no stored SQ counters of this build for this workload: HBM figures only; the path is bound by fp64 VALU issue latency, not by HBM (SURVEY 8d)"}
    blk.update(common)
    if second:
        blk["second_kernel"] = second
    return blk


if __name__ == "__main__":
    sys.exit(main() or 0)
