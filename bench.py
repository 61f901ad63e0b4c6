#!/usr/bin/env python3
"""Headline benchmark: composite-llk evaluations/s over a (split x mi-rate) grid at
128 merged PSMC intervals (BASELINE.json `metric`, configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

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
# one-to-one onto hardware queues up to 24 queues per process; one queue more and the driver time-slices
# them (throughput collapses 2x).  22 lanes is the single-process peak; defaults leave room for the null
# stream and, with several ranks, for RCCL's and torch's own streams.
# (single rank with RCCL initialised and one all_gather per 8 batches of a lane: 16 lanes 2.76e7, 18 lanes 2.86e7,
# 20 lanes 2.65e7, 22 lanes 2.80e7 evals/s against 2.97e7 without RCCL -> 18 with several ranks)
DEFAULT_STREAMS = 20 if int(os.environ.get("WORLD_SIZE", "1")) <= 1 else 18
HW_QUEUES = 24
HW_QUEUE_SLACK = 4


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
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(q if q else max(HW_QUEUES, n + HW_QUEUE_SLACK)))


_early_streams()

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="config2", help="config2 (headline) | config3 | config4 | config5 | config3-search")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak: every rank its own grid; strong: ONE grid sharded over the ranks (config4 / config5)")
    ap.add_argument("--min-seconds", type=float, default=0.25, help="repeat the K-step timed loop until the timed regions add up to this; the median is reported")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample (wall)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the all_gather path even with one rank (rehearsal)")
    ap.add_argument("--gather-bucket", type=int, default=8,
                    help="multi-GPU: batches of a lane whose llk share ONE all_gather (fewer, larger collectives: xGMI is latency-bound at 32 KB)")
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES (default: %d)" % HW_QUEUES)
    ap.add_argument("--streams", type=int, default=DEFAULT_STREAMS,
                    help="HIP streams the steps are issued on round-robin: independent batches overlap (1 = strictly serial)")
    return ap.parse_args()


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


def algorithmic_bytes(w, n_rep, idx=None):
    """HBM bytes the algorithm needs per launch, per kernel.

    correction: one chain per distinct parameter vector, computed up to the largest split index of
                its members: 8P in; per interval 16 B of rates + 48 B of pair state out (+ the trunk records,
                1 056 B per chain and interval, when the trunk wave follows its chain in the same launch);
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
        trunk = sum(1056 * (L + 1) for L in chains.values())
        spectrum += 1056 * n                                  # one trunk record per candidate
        if len(chains) <= 256 or n <= 2048:                   # one chain per wave: the trunk wave follows its chain inside the chain launch
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


def main():
    a = parse()
    # stdout carries exactly one JSON line: libraries that print banners there (RCCL prints its version
    # block on communicator creation) are sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from misti_amd import workloads
    from misti_amd.dist import env_rank, shard_indices
    from misti_amd.engine import Engine, truth_spectrum

    rank, local_rank, world = env_rank()
    if a.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    if a.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
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

    # ---- workload ---------------------------------------------------------------
    spec = lambda *args: truth_spectrum(*args, device=local_rank)
    if a.workload == "config3-search":
        from misti_amd import search_bench
        return search_bench.run(a, spec, dev, local_rank, rank, world, json_fd)
    w = workloads.BUILDERS[a.workload](spec)
    strong = a.scaling == "strong"
    if strong:
        mine = shard_indices(w.n_cand, rank, world, interleave=True)      # ONE grid, interleaved shards
    else:
        mine = np.arange(w.n_cand)
        if world > 1 and w.params is not None:
            # distinct grids per rank (weak scaling): shift the rate axis by a rank-dependent factor
            w.params = w.params * (1.0 + 0.01 * rank)
    n_total = w.n_cand
    n, R, P = len(mine), int(w.jsfs.shape[0]), w.n_param
    per = -(-n_total // world) if strong else n                             # rows every rank contributes to the gather
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
        def __init__(self):
            self.eng = Engine(w.times, w.lh, device=local_rank, **w.engine_kwargs())
            self.stream = torch.cuda.ExternalStream(self.eng.stream_handle(), device=dev)
            # multi-GPU: the llk of `bucket` consecutive batches of this lane are gathered by ONE collective (a 32 KB all_gather
            # per batch is pure latency on xGMI and costs a fifth of the rate); every batch's llk still reaches every rank
            self.slots = torch.full((bucket, per, R), float("nan"), dtype=torch.float64, device=dev)   # rows beyond n: padding of a ragged shard
            self.gathered = torch.empty((world * bucket * per, R), dtype=torch.float64, device=dev) if use_dist else None
            self.fill = 0
            self.llk = self.slots[0]
            self.jafs = torch.empty((n, 7), dtype=torch.float64, device=dev)
            self.status = torch.empty(n, dtype=torch.int32, device=dev)

        def step(self):
            self.llk = self.slots[self.fill]
            self.eng.evaluate_dev(n, d_split.data_ptr(), d_par.data_ptr() if P else 0, R, d_jsfs.data_ptr(),
                                  self.llk.data_ptr(), self.jafs.data_ptr(), 0, 0, self.status.data_ptr())
            if use_dist:
                self.fill += 1
                if self.fill == bucket:
                    self.flush()

        def flush(self):
            """The bucket's collective, issued from the lane's own stream: c10d orders it after the batches (event on this
            stream), runs it on its communicator stream in host issue order - the same on every rank - and makes this
            stream wait for it, so the next batch of the lane cannot overwrite a slot early.  (A dedicated communication
            stream + events per step cost a hardware queue and 15 % of the rate.)  A partly filled bucket is gathered whole."""
            if use_dist and self.fill > 0:
                with torch.cuda.stream(self.stream):
                    dist.all_gather_into_tensor(self.gathered, self.slots.view(bucket * per, R))
                self.fill = 0

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    host_issue = [0.0]

    def timed(lanes, steps):
        """EXACTLY `steps` steps issued round-robin on `lanes` between two fences; seconds (max over ranks)."""
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            lanes[i % len(lanes)].step()
        for lane in lanes:                           # what is still in a bucket belongs to the timed steps
            lane.flush()
        host_issue[0] = time.perf_counter() - t0
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def repeated(lanes, steps, warmup, min_seconds):
        """W untimed warmup steps, then the K-step timed region again and again until the regions add up to
        min_seconds (at least once, at most 200 times; every rank takes the same decision: dt is the all-reduced max)."""
        for i in range(warmup):
            lanes[i % len(lanes)].step()
        for lane in lanes:
            lane.flush()
        dts = []
        while not dts or (sum(dts) < min_seconds and len(dts) < 200):
            dts.append(timed(lanes, steps))
        return dts

    n_streams = max(1, a.streams)
    main_lanes = [Lane() for i in range(n_streams)]
    # context initialisation, not measurement: the first batch of a context allocates its workspaces (hipMalloc) and
    # learns its launch shape; every lane does that once here so that a short run (K < lanes x a few) times steady state
    for lane in main_lanes:
        lane.step()
    fence()
    dts = repeated(main_lanes, a.steps, a.warmup, a.min_seconds)
    dt = statistics.median(dts)
    issue_main = host_issue[0]
    # strictly serial pass on one stream: the per-grid latency, and the per-kernel durations (HIP events on the launch stream)
    serial = main_lanes[0]
    eng = serial.eng
    k_serial = max(4, min(a.steps, 16))
    dts_serial = repeated([serial], k_serial, 2, a.min_seconds)
    dt_serial = statistics.median(dts_serial)
    eng.enable_timing(True)
    eng.kernel_times(reset=True)
    timed([serial], k_serial)
    kms, kn = eng.kernel_times(reset=True)
    eng.enable_timing(False)

    status = serial.status.cpu().numpy()
    llk = serial.llk[:n].cpu().numpy()
    job_cands = n_total if strong else world * n          # candidates all ranks evaluate per step
    evals = job_cands * R * a.steps
    value = evals / dt
    chains_mine = len(chain_lengths(w, mine))
    if use_dist:
        t = torch.zeros(world, dtype=torch.int64, device=dev)
        t[rank] = chains_mine
        dist.all_reduce(t)
        chains_per_rank = [int(v) for v in t.cpu()]
        cand_t = torch.zeros(world, dtype=torch.int64, device=dev)
        cand_t[rank] = n
        dist.all_reduce(cand_t)
        cands_per_rank = [int(v) for v in cand_t.cpu()]
    else:
        chains_per_rank, cands_per_rank = [chains_mine], [n]

    metric = "composite-llk evals/sec over (split×mi) grid, 128 merged PSMC intervals"
    try:                                              # the exact string of BASELINE.json when it is at hand
        metric = json.load(open(os.path.join(ROOT, "BASELINE.json"))).get("metric") or metric
    except Exception:
        pass
    out = {
        "metric": metric,
        "value": value, "unit": "llk evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": w.name, "candidates_per_gpu": n, "candidates_total": job_cands, "replicates": R, "numT": w.numT,
                   "batches_in_flight": n_streams, "streams": n_streams,
                   "world_size": group_world, "candidates_per_rank": cands_per_rank, "chains_per_rank": chains_per_rank,
                   "gather_bucket": bucket if use_dist else None,
                   "parallelism": ("ONE grid sharded over the ranks (interleaved), all_gather of llk (RCCL), one collective per %d batches of a lane" % bucket if strong else
                                   "every rank its own grid, all_gather of llk (RCCL), one collective per %d batches of a lane" % bucket) if world > 1 else "1 GPU"},
        "timing": {"repeats": len(dts), "timed_region_s_median": dt, "timed_region_s_total": sum(dts),
                   "timed_region_s_min": min(dts), "timed_region_s_max": max(dts),
                   "note": "value and ms_per_step are the MEDIAN repetition of the K-step timed region (each between barriers, max over ranks)"},
    }
    out_spectra_per_s = job_cands * a.steps / dt
    if rank == 0:
        out["single_batch"] = {"value": job_cands * R * k_serial / dt_serial, "ms_per_step": 1e3 * dt_serial / k_serial, "steps": k_serial,
                               "streams": 1, "repeats": len(dts_serial),
                               "note": "the same step issued strictly one after another on one stream: one grid's latency, bounded by its "
                                       "longest lambda-correction chain (up to ~900 dependent residual evaluations in the runaway corner of "
                                       "the grid); `value` above overlaps config.batches_in_flight such batches"}
        out["serial"] = out["single_batch"]          # round-1 name of the same block
        # ---- roofline of the dominant kernel -----------------------------------------
        per_ms = {k: (kms[k] / kn[k] if kn[k] else 0.0) for k in kms}          # ms per launch, HIP events on the launch stream
        dom = max(("correct", "spectrum"), key=lambda k: per_ms[k])
        ab = algorithmic_bytes(w, R, mine)
        achieved = ab[dom] / (per_ms[dom] * 1e-3) / 1e9 if per_ms[dom] > 0 else 0.0
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc) and world == 1:
            try:
                j = json.load(open(pmc))
                if j.get("workload") == a.workload:
                    traffic = j.get(dom + "_hbm_bytes_per_launch")
                    traffic_source = "stored: profiles/pmc_latest.json <- " + str(j.get("source")) + " (not measured in this run)"
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "kernel": dom + "_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                           "algorithmic_bytes_per_launch": ab[dom], "candidates_per_launch": n, "chains_per_launch": ab["n_chains"],
                           "ms_per_launch": per_ms,
                           "note": "the path is neither HBM- nor MFMA-bound (SURVEY 8d): ~1 KB per candidate against ~1e5 dependent fp64 "
                                   "operations; the correction kernel is bound by the dependent-issue latency of its longest chain "
                                   "(serial trust-region iterations of the reference's solver), the spectrum kernel by fp64 VALU issue + LDS latency"}
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
        out["host_issue_ms_per_step"] = 1e3 * issue_main / a.steps
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


if __name__ == "__main__":
    main()
