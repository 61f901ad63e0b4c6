#!/usr/bin/env python3
"""Command line with the option surface of the reference's ``MiSTI.py``
(``/root/reference/MiSTI.py:43-260``) on top of the HIP engine.

    python -m misti_amd.cli g1.psmc g2.psmc data.sfs 64 -mi 1 4 64 0.2 1 --cpfit -uf

Same positional arguments and options (``-o -wd -tol -mth -mi -pu --sdate --hetloss
--discr -rd --funits -uf --nosmooth --trueEPS --cpfit -bs --debug``), same printed
result line (``bs_id = ... splitT = ... time = ... migration rates ... llh = ...``,
MiSTI.py:240 - what the ``test.bs`` scripts grep), ``-o`` written only for ``-bs 0``
(MiSTI.py:248-249).  ``--psmcMode 1`` (experimental PSMC re-estimation) is not offered.

Batched extension (no reference counterpart; replaces the bash loops of ``test.bs/*.sh``
and the GNU-parallel recipe of ``README.md:110-115``):

    --grid-st A B [STEP]      scan split times A..B (inclusive), bands ending at the
                              given split time follow each candidate's split
    --grid-mi K LO HI N       N log-spaced values for the K-th optimised parameter
    --all-bs                  evaluate every row of the JSFS file as a replicate
    --gpus N                  the sweep on N GPUs of the node: this process starts N ranks (one per GPU, torch.distributed over
                              RCCL), whole lambda-correction chains are dealt to the ranks, one all_gather, rank 0 prints
    --devices 0,1,...         the sweep on a LIST of devices from this one process (misti_create_multi: one context and one host
                              thread per entry)
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from math import ceil

import numpy as np

from . import io as mio
from .engine import BatchResult, Engine, MigrationInference


RANK_MODULE = "misti_amd.cli"       # what `--gpus N` starts N times (a test driver that wraps this module names itself here)


def build_parser():
    p = argparse.ArgumentParser(description="Migration inference from PSMC (MI355X engine).")
    p.add_argument("fpsmc1", help="psmc file 1")
    p.add_argument("fpsmc2", help="psmc file 2")
    p.add_argument("fjafs", help="joint allele frequency spectrum file")
    p.add_argument("st", type=float, help="split time")
    p.add_argument("-o", "--fout", default="", help="output file, default is stdout")
    p.add_argument("-wd", default="", help="working directory (path to data files)")
    p.add_argument("-tol", type=float, default=1e-4, help="optimisation precision (default is 1e-4)")
    p.add_argument("-mth", type=float, default=0.0, help="mixture treshhold (default is 0.0)")
    p.add_argument("-mi", nargs=5, action="append", default=[],
                   help="migration rate: source population (1 or 2), start, end, initial value, fixed(0)/optimised(1)")
    p.add_argument("-pu", nargs=4, action="append", default=[],
                   help="pulse migration: source population (1 or 2), time, rate, fixed(0)/optimised(1)")
    p.add_argument("--sdate", type=float, default=0, help="dating of the second sample (for ancient genome)")
    p.add_argument("--hetloss", "-hl", nargs=2, type=float, help="loss of heterozygosity for the two genomes")
    p.add_argument("--discr", "-d", type=int, default=1, help="accepted for compatibility (the reference ignores it)")
    p.add_argument("-rd", type=int, default=-1, help="round (RD) in the PSMC files, -1 = last")
    p.add_argument("--funits", type=str, default="setunits.txt", help="file with units to rescale times and EPS")
    p.add_argument("-uf", action="store_true", help="unfolded spectrum")
    p.add_argument("--nosmooth", action="store_true", help="don't smooth")
    p.add_argument("--trueEPS", action="store_true", help="input is true effective population size")
    p.add_argument("--cpfit", action="store_true", help="fit probabilities to coalesce within each interval")
    p.add_argument("--bsMode", "-bs", type=int, default=-1, help="use JSFS row N (-1: sum of all rows)")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--device", type=int, default=0, help="HIP device index")
    p.add_argument("--grid-st", nargs="+", type=float, metavar="V", help="A B [STEP]: scan split times")
    p.add_argument("--grid-mi", nargs=4, action="append", default=[], metavar=("K", "LO", "HI", "N"),
                   help="log-spaced values for optimised parameter K")
    p.add_argument("--all-bs", action="store_true", help="evaluate every JSFS row as a bootstrap replicate")
    p.add_argument("--gpus", type=int, default=1, help="grid mode: start this many ranks, one per GPU (replaces `parallel -j N ./MiSTI.py ...`)")
    p.add_argument("--devices", type=str, default="", help="grid mode: comma-separated device list evaluated from this one process (misti_create_multi)")
    return p


def _evaluator(a, inp, bands, pulses, k, device):
    """The batch evaluator of grid mode: ``evaluate(split, params, rows)`` -> an object with ``llk[n][R]`` and ``status[n]``, and a
    function closing it: the HIP engine on ``device``, or on the device list of ``--devices`` (the multi-device C ABI).  There is no
    other evaluator: without a usable GPU the constructor fails."""
    flags = dict(cpfit=a.cpfit, true_eps=a.trueEPS, smooth=not a.nosmooth, unfolded=a.uf)
    if a.devices:
        from .engine import MultiEngine
        e = MultiEngine(inp.times, inp.lambdas, bands, pulses, n_param=k, sample_date=inp.sampleDateDiscr, mixture_th=a.mth,
                        devices=[int(d) for d in a.devices.split(",")], **flags)
    else:
        e = Engine(inp.times, inp.lambdas, bands, pulses, n_param=k, sample_date=inp.sampleDateDiscr, mixture_th=a.mth, device=device, **flags)
    return e.evaluate, e.close


def grid_mode(a, inp, rows):
    """Batched sweep: one Engine, candidates = split values x parameter grid, replicates = JSFS rows."""
    st0 = a.st
    splits = [st0]
    if a.grid_st:
        lo, hi = a.grid_st[0], a.grid_st[1]
        step = a.grid_st[2] if len(a.grid_st) > 2 else 1.0
        splits = list(np.arange(lo, hi + 0.5 * step, step))
    bands, pulses, k = [], [], 0
    init = []
    for el in a.mi:
        pop, start, end, val, opt = int(el[0]) - 1, int(el[1]), int(el[2]), float(el[3]), int(el[4])
        if a.grid_st and end == int(ceil(st0)):
            end = -1                                  # follows the candidate's split (test.bs/san_sar.bs.sh:36)
        bands.append((pop, start, end, val, k if opt else -1))
        if opt:
            init.append(val)
            k += 1
    for el in a.pu:
        pop, t, val, opt = int(el[0]) - 1, int(el[1]), float(el[2]), int(el[3])
        pulses.append((pop, t, val, k if opt else -1))
        if opt:
            init.append(val)
            k += 1
    axes = [np.array([v]) for v in init]
    for g in a.grid_mi:
        axes[int(g[0])] = np.logspace(np.log10(float(g[1])), np.log10(float(g[2])), int(g[3]))
    mesh = np.meshgrid(np.array(splits), *axes, indexing="ij")
    split = mesh[0].ravel()
    params = np.stack([m.ravel() for m in mesh[1:]], axis=1) if k else None
    data = np.array(rows if a.all_bs else [rows[a.bsMode] if a.bsMode >= 0 else np.sum(rows, axis=0)], dtype=float)
    # --gpus N: this process is one of N ranks (main() started them); whole chains per rank, one all_gather, rank 0 prints
    from . import dist as mdist
    rank, local, world = mdist.init_from_env()
    t0 = time.time()
    evaluate, close = _evaluator(a, inp, bands, pulses, k, local if world > 1 else a.device)
    try:
        if world > 1:
            llk, status = mdist.evaluate_sharded(evaluate, split, params, data, by_chain=True, with_status=True)
            res = BatchResult(llk.cpu().numpy(), None, status.cpu().numpy())
        else:
            res = evaluate(split, params, data)
    finally:
        close()
    dt = time.time() - t0
    if "WORLD_SIZE" in os.environ:                    # started as a rank (by --gpus N, or by torchrun directly): leave the group in order
        import torch.distributed as tdist
        if tdist.is_initialized():
            tdist.barrier()
            tdist.destroy_process_group()
    if world > 1:
        if rank != 0:
            return 0
        print("Sharded over %d ranks (whole chains per rank; one all_gather of %d x %d log-likelihoods)" % (world, len(split), data.shape[0]))
    for c in range(len(split)):
        pstr = "" if params is None else "\t".join("%.6g" % v for v in params[c])
        for r in range(data.shape[0]):
            print("bs_id =", r if a.all_bs else a.bsMode, "\tsplitT =", split[c], "\tparams", pstr, "\tllh =", res.llk[c, r],
                  "\tstatus =", int(res.status[c]))
    best = np.unravel_index(np.argmax(np.where(np.isfinite(res.llk), res.llk, -np.inf)), res.llk.shape)
    print("\nbest: splitT =", split[best[0]], "params =", None if params is None else list(params[best[0]]),
          "replicate =", best[1], "llh =", res.llk[best])
    if a.all_bs and len(splits) > 1 and data.shape[0] > 1:
        # the bootstrap confidence interval of test.bs/bs_conf_int.ipynb: per replicate the split of the best candidate,
        # then a Student-t interval of those maxima (row 0 of a -bs file is the sum of the chunks, as there)
        from .optimize import bootstrap_split_interval
        mean, (lo, hi), best_split = bootstrap_split_interval(res.llk, split)
        print("bootstrap: best splitT per replicate mean = %.6g, 95%% interval = [%.6g, %.6g] over %d replicates"
              % (mean, lo, hi, data.shape[0]))
    print("Evaluated %d candidates x %d replicates in %.3f s (%.0f llk evals/s); %.1f%% without a value"
          % (len(split), data.shape[0], dt, res.llk.size / dt, 100 * res.fraction_failed))
    return 0


def main(argv=None):
    t0 = time.time()
    a = build_parser().parse_args(argv)
    # the two ways of using several GPUs exclude each other: N ranks that each opened the whole device list would run N x D contexts
    if a.gpus > 1 and a.devices:
        print("--gpus (one rank per GPU) and --devices (a device list in one process) exclude each other", file=sys.stderr)
        return 2
    if a.devices and not (a.grid_st or a.grid_mi or a.all_bs):
        print("--devices applies to the batched sweep (--grid-st / --grid-mi / --all-bs); a single model runs on --device", file=sys.stderr)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if not (a.grid_st or a.grid_mi or a.all_bs):
            print("--gpus applies to the batched sweep (--grid-st / --grid-mi / --all-bs); a single model runs on one GPU", file=sys.stderr)
        else:
            # Start the ranks as CHILD processes and pass rank 0's output through.  This process has not imported torch or touched
            # HIP and never does (a process that has initialised the GPU must not be forked or replaced); the reference's way of
            # using N processors is `parallel -j N ./MiSTI.py ... >> res.out` (/root/reference/README.md:110-115).
            from . import dist as mdist
            code, out = mdist.launch_ranks(a.gpus, list(sys.argv[1:] if argv is None else argv), module=RANK_MODULE)
            sys.stdout.write(out)
            sys.stdout.flush()
            return code
    quiet = int(os.environ.get("RANK", "0")) != 0           # a rank other than 0 of a --gpus run: it computes, rank 0 reports
    if quiet:
        sys.stdout = open(os.devnull, "w")
    units = mio.Units.from_file(a.funits)
    print(units.describe())
    if a.hetloss is not None:
        units.set_hetloss(a.hetloss)
    print(" ".join(sys.argv))
    print(time.strftime("Job run at %H:%M:%S on %d %b %Y"))
    f1, f2, fj = (os.path.join(a.wd, f) for f in (a.fpsmc1, a.fpsmc2, a.fjafs))
    print("Reading from files:")
    print("pop1\t", f1)
    print("pop2\t", f2)
    print("jafs\t", fj)
    rows, pop1, pop2 = mio.read_jsfs(fj)
    if a.bsMode == -1:
        inputSFS = [sum(r[i] for r in rows) for i in range(8)]
    else:
        inputSFS = rows[a.bsMode]
    print("IMPORTANT NOTICE!!! Every time you are running MiSTI, make sure that psmc files are supplied in the same "
          "order as populations appear in the joint allele frequency spectrum.")
    fout = os.path.join(a.wd, a.fout) if a.fout else ""
    inp = mio.read_psmc(f1, f2, a.sdate, a.rd, units)
    inp.divergenceTime = a.st
    if a.grid_st or a.grid_mi or a.all_bs:
        return grid_mode(a, inp, rows)

    t1 = time.time()
    mig = MigrationInference(inp.times, inp.lambdas, inputSFS, inp.divergenceTime, a.mi, a.pu,
                             thrh=[inp.theta, inp.rho], Tpsmc=inp.Tpsmc, enableOutput=False, smooth=not a.nosmooth,
                             unfolded=a.uf, trueEPS=a.trueEPS, sampleDate=inp.sampleDateDiscr, mixtureTH=a.mth,
                             cpfit=a.cpfit, device=a.device)
    sol = mig.Solve(a.tol)
    print(sol)
    print("\nParameter estimates:")
    fixed = [float(el[3]) for el in a.mi if int(el[4]) == 0]
    fixed_s = "fixed = [" + ", ".join(str(v) for v in fixed) + "]" if fixed else ""
    opt_s = "optim = [" + ", ".join(str(v) for v in sol[0]) + "]" if len(sol[0]) > 0 else ""
    mig_s = fixed_s + "\t" + opt_s if fixed_s and opt_s else fixed_s + opt_s
    # inp.times was extended in place by a fractional split, as in the reference (MiSTI.py:240)
    print("bs_id =", a.bsMode, "\tsplitT =", inp.divergenceTime, "\ttime =",
          sum(inp.times[0:ceil(inp.divergenceTime)]) * inp.scaleTime, "\tmigration rates", mig_s, "\tllh =", sol[1])
    print("\n")
    t2 = time.time()
    if sol[1] == -10 ** 9:
        print("Failed to fit such a model.")
    elif a.bsMode == 0:
        llh = mig.llh if len(sol[0]) == 0 else mig.JAFSLikelihood(sol[0])
        text = mio.format_migration(mig, llh, inp.scaleTime, inp.scaleEPS)
        if fout == "":
            print(text)
        else:
            with open(fout, "w") as fw:
                fw.write(text)
    MigrationInference.Report()
    print("Runtime:   optimisation", t2 - t1)
    print("           total       ", time.time() - t0)
    return 0


if __name__ == "__main__":
    sys.exit(main())
