"""The configurations of BASELINE.json as concrete synthetic workloads.

Every builder returns a ``Workload``: the merged grid, PSMC-like rates derived
from a true model (so that the lambda-correction has a solution near the truth),
band / pulse descriptors in C-ABI form, flags, and the candidate batch.  The data
JSFS needs the expected spectrum at the truth; the caller supplies
``spectrum_fn(times, lh, split, bands, pulses, sample_date) -> jafs[7]`` (the HIP
engine in ``trueEPS`` mode on the GPU box, the oracle in CPU tests).
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field

import numpy as np

from . import io as mio
from . import synth


@dataclass
class Workload:
    name: str
    times: list
    lh: list
    bands: list                 # (pop, start, end(-1 = split), value, param)
    pulses: list                # (pop, time, value, param)
    n_param: int
    flags: dict                 # cpfit / true_eps / smooth / unfolded
    sample_date: int
    split_time: np.ndarray      # [n_cand]
    params: np.ndarray | None   # [n_cand][n_param]
    truth: dict = field(default_factory=dict)
    jsfs: np.ndarray | None = None   # [n_rep][8]

    @property
    def n_cand(self):
        return int(self.split_time.shape[0])

    @property
    def numT(self):
        return len(self.lh)

    def engine_kwargs(self):
        return dict(bands=self.bands, pulses=self.pulses, n_param=self.n_param, sample_date=self.sample_date, **self.flags)


def _truth_bands(bands, split):
    return [(p, s, split if e < 0 else e, v, -1) for p, s, e, v, _ in bands]


def _mis_pus(bands, pulses, split):
    mis = [[p + 1, s, split if e < 0 else e, v, 0] for p, s, e, v, _ in bands]
    pus = [[p + 1, t, v, 0] for p, t, v, _ in pulses]
    return mis, pus


def config1(spectrum_fn, n_sites=10 ** 6):
    """numT = 32, split 20, no migration, a single candidate (plumbing case)."""
    inp = synth.psmc_pair(16, 17)
    times, lh, _ = synth.self_consistent(inp, 20)
    jafs = spectrum_fn(times, lh, 20, [], [], 0)
    w = Workload("config1: numT=32 split=20 no migration", times, lh, [], [], 0,
                 dict(cpfit=False, true_eps=False, smooth=True, unfolded=False), 0,
                 np.array([20.0]), None, dict(split=20))
    w.jsfs = np.array([synth.counts_from_spectrum(jafs, n_sites)])
    return w


def config2(spectrum_fn, n_split=64, n_rate=64, first_split=32, true_split=64, true_rate=0.2, n_sites=10 ** 6, max_rate=1.0, cpfit=True,
            psmc_seeds=(1, 2), truth_seed=7, psmc_rows=(64, 65)):
    """numT = 128; grid of split index x rate of one band ``-mi 1 4 {st} {r} 1``, ``--cpfit`` (``cpfit=False``: the
    reference's default fit, MiSTI.py:86,213)."""
    inp = synth.psmc_pair(psmc_rows[0], psmc_rows[1], seeds=psmc_seeds)
    band_truth = [(0, 4, true_split, true_rate, -1)]
    mis, pus = _mis_pus(band_truth, [], true_split)
    times, lh, _ = synth.self_consistent(inp, true_split, mis, pus, seed=truth_seed)
    jafs = spectrum_fn(times, lh, true_split, band_truth, [], 0)
    splits = np.arange(first_split, first_split + n_split, dtype=np.float64)
    rates = np.logspace(-3, np.log10(max_rate), n_rate)
    st, rr = np.meshgrid(splits, rates, indexing="ij")
    w = Workload("config2: numT=128, %dx%d split x mi-rate grid, one band, %s" % (n_split, n_rate, "--cpfit" if cpfit else "default fit"),
                 times, lh, [(0, 4, -1, 0.0, 0)], [], 1,
                 dict(cpfit=cpfit, true_eps=False, smooth=True, unfolded=False), 0,
                 st.ravel().copy(), rr.ravel()[:, None].copy(), dict(split=true_split, rate=true_rate))
    w.jsfs = np.array([synth.counts_from_spectrum(jafs, n_sites)])
    return w


def config2b(spectrum_fn, cpfit=True):
    """The headline grid's shape on OTHER data: other PSMC curves (seeds 3, 4), another true history (seed 8), true split 70, true rate 0.1, the
    split axis 40 ... 103.  Made at the end of round 5 as a held-out instance: nothing in the kernels, the contract or the tests was tuned on it
    (tools/fullsize_report.py config2b config2b:default; tests/golden/make_fullsize.py)."""
    w = config2(spectrum_fn, first_split=40, true_split=70, true_rate=0.1, cpfit=cpfit, psmc_seeds=(3, 4), truth_seed=8)
    w.name = "config2b: held-out instance of " + w.name
    return w


def config2c(spectrum_fn, cpfit=True):
    """A second held-out instance, made AFTER the stall rule that config2b led to (PSMC seeds 5, 6; true history seed 9; true split 60, true rate 0.3; split
    axis 36 ... 99): does what was learnt on config2b hold on data nobody has looked at?"""
    w = config2(spectrum_fn, first_split=36, true_split=60, true_rate=0.3, cpfit=cpfit, psmc_seeds=(5, 6), truth_seed=9)
    w.name = "config2c: second held-out instance of " + w.name
    return w


def config2t(spectrum_fn, cpfit=False):
    """The held-out grid config2b with --trueEPS (MigrationInference.py:74, :325-326: no lambda-correction; the spectrum path alone at full size)."""
    w = config2b(spectrum_fn, cpfit=False)
    w.flags = dict(w.flags, true_eps=True)
    w.name = "config2t: --trueEPS on " + w.name
    return w


def config2u(spectrum_fn, cpfit=True):
    """The held-out grid config2b with the UNFOLDED spectrum (--uf, MigrationInference.py:600-609: seven classes instead of four folded ones), no smoothing (--nosmooth)."""
    w = config2b(spectrum_fn, cpfit=cpfit)
    w.flags = dict(w.flags, unfolded=True, smooth=False)
    w.name = "config2u: unfolded, no smoothing, on " + w.name
    return w


def config2m(spectrum_fn, cpfit=True):
    """Held-out grid with the band in the OTHER direction (``-mi 2 6 {st} {r} 1``: population 2, from interval 6) on yet other data (PSMC seeds 11, 12; history seed 13;
    true split 66, true rate 0.12)."""
    inp = synth.psmc_pair(64, 65, seeds=(11, 12))
    true_split, true_rate = 66, 0.12
    band_truth = [(1, 6, true_split, true_rate, -1)]
    mis, pus = _mis_pus(band_truth, [], true_split)
    times, lh, _ = synth.self_consistent(inp, true_split, mis, pus, seed=13)
    jafs = spectrum_fn(times, lh, true_split, band_truth, [], 0)
    splits = np.arange(36, 100, dtype=np.float64)
    rates = np.logspace(-3, 0, 64)
    st, rr = np.meshgrid(splits, rates, indexing="ij")
    w = Workload("config2m: held-out, numT=128, 64x64 split x mi-rate grid, one band into population 2, %s" % ("--cpfit" if cpfit else "default fit"),
                 times, lh, [(1, 6, -1, 0.0, 0)], [], 1, dict(cpfit=cpfit, true_eps=False, smooth=True, unfolded=False), 0,
                 st.ravel().copy(), rr.ravel()[:, None].copy(), dict(split=true_split, rate=true_rate))
    w.jsfs = np.array([synth.counts_from_spectrum(jafs, 10 ** 6)])
    return w


def config2f(spectrum_fn, cpfit=True):
    """The held-out grid config2b with FRACTIONAL split times (MigrationInference.py:89-99: the split inside an interval): the split axis 40.3, 41.05, 41.8, ... (step 0.75)."""
    w = config2b(spectrum_fn, cpfit=cpfit)
    n = w.n_cand // 64
    splits = 40.3 + 0.75 * np.arange(n)
    w.split_time = np.repeat(splits, 64)
    w.name = "config2f: fractional split times on " + w.name
    return w


def config2pu(spectrum_fn, cpfit=True):
    """Held-out grid of a PULSE-only model (``-pu 2 30 {f} 1`` and a fixed band ``-mi 1 8 {st} 0.05``) on yet other data (PSMC seeds 13, 14; history seed 15; true split 72,
    true pulse 0.2): 64 split values x 64 pulse fractions 0 ... 0.6."""
    inp = synth.psmc_pair(64, 65, seeds=(13, 14))
    true_split = 72
    band_truth = [(0, 8, true_split, 0.05, -1)]
    pulse_truth = [(1, 30, 0.2, -1)]
    mis, pus = _mis_pus(band_truth, pulse_truth, true_split)
    times, lh, _ = synth.self_consistent(inp, true_split, mis, pus, seed=15)
    jafs = spectrum_fn(times, lh, true_split, band_truth, pulse_truth, 0)
    splits = np.arange(40, 104, dtype=np.float64)
    fr = np.linspace(0.0, 0.6, 64)
    st, ff = np.meshgrid(splits, fr, indexing="ij")
    w = Workload("config2pu: held-out, numT=128, 64x64 split x pulse-fraction grid, fixed band, %s" % ("--cpfit" if cpfit else "default fit"),
                 times, lh, [(0, 8, -1, 0.05, -1)], [(1, 30, 0.0, 0)], 1, dict(cpfit=cpfit, true_eps=False, smooth=True, unfolded=False), 0,
                 st.ravel().copy(), ff.ravel()[:, None].copy(), dict(split=true_split, pulse=0.2))
    w.jsfs = np.array([synth.counts_from_spectrum(jafs, 10 ** 6)])
    return w


def config2n64(spectrum_fn, cpfit=True):
    """Held-out instance at another grid size: numT = 64 (PSMC files of 32 and 33 rows, seeds 7, 8; true history seed 10; true split 30, rate 0.15), 32 splits x 64 rates."""
    w = config2(spectrum_fn, n_split=32, first_split=16, true_split=30, true_rate=0.15, cpfit=cpfit, psmc_seeds=(7, 8), truth_seed=10, psmc_rows=(32, 33))
    w.name = "config2n64: held-out, numT=64: " + w.name
    return w


def config2n255(spectrum_fn, cpfit=True):
    """Held-out instance at the largest grid: numT = 255 (two PSMC files of 128 rows, seeds 9, 10; true history seed 11; true split 120, rate 0.08), 64 splits x 64 rates."""
    w = config2(spectrum_fn, first_split=70, true_split=120, true_rate=0.08, cpfit=cpfit, psmc_seeds=(9, 10), truth_seed=11, psmc_rows=(128, 128))
    w.name = "config2n255: held-out, numT=255: " + w.name
    return w


def config3b(spectrum_fn, cpfit=True):
    """Held-out instance of config 3 (end of round 5, after the stall rule): other PSMC curves (seeds 3, 4), another true history (seed 8), other random starts (seed 6)."""
    w = config3(spectrum_fn, seed=6, cpfit=cpfit, psmc_seeds=(3, 4), truth_seed=8)
    w.name = "config3b: held-out instance of " + w.name
    return w


def config4b(spectrum_fn, cpfit=False):
    """Held-out instance of config 4 (end of round 5): other PSMC curves (seeds 3, 4), another true history (seed 8), true split 61, another bootstrap table (seed 4)."""
    w = config4(spectrum_fn, true_split=61, cpfit=cpfit, seed=4, psmc_seeds=(3, 4), truth_seed=8)
    w.name = "config4b: held-out instance of " + w.name
    return w


def config5b(spectrum_fn, cpfit=True):
    """Held-out instance of config 5 (end of round 5, after the stall rule): other PSMC curves (seeds 3, 4), another true history (seed 8)."""
    w = config5(spectrum_fn, cpfit=cpfit, psmc_seeds=(3, 4), truth_seed=8)
    w.name = "config5b: held-out instance of " + w.name
    return w


def config2x16(spectrum_fn, n_grid=16, **kw):
    """``n_grid`` config-2 grids with distinct rate axes in ONE batch (the README's ``st x mc x rates`` sweep as a caller
    with one large sweep issues it, /root/reference/README.md:113): grid g scales the rate axis by 1 + g/64, so every
    (grid, rate) pair is its own chain - 65 536 candidates in 1 024 chains for the default 16 x (64 x 64)."""
    w = config2(spectrum_fn, **kw)
    split = np.tile(w.split_time, n_grid)
    par = np.concatenate([w.params * (1.0 + g / 64.0) for g in range(n_grid)], axis=0)
    w.name = "config2x%d: %d config-2 grids with distinct rate axes in one batch (%d candidates), --cpfit" % (n_grid, n_grid, split.shape[0])
    w.split_time, w.params = split, par
    return w


def config3(spectrum_fn, n_start=16384, true_split=64, seed=5, n_sites=10 ** 6, cpfit=True, psmc_seeds=(1, 2), truth_seed=7):
    """Two optimised bands, random starts (one batched simplex-vertex evaluation)."""
    inp = synth.psmc_pair(64, 65, seeds=psmc_seeds)
    band_truth = [(0, 4, true_split, 0.2, -1), (1, 10, true_split, 0.05, -1)]
    mis, pus = _mis_pus(band_truth, [], true_split)
    times, lh, _ = synth.self_consistent(inp, true_split, mis, pus, seed=truth_seed)
    jafs = spectrum_fn(times, lh, true_split, band_truth, [], 0)
    rng = np.random.default_rng(seed)
    par = 10.0 ** rng.uniform(-3, 0, size=(n_start, 2))
    w = Workload("config3: numT=128, two optimised bands, %d random starts, %s" % (n_start, "--cpfit" if cpfit else "default fit"),
                 times, lh, [(0, 4, true_split, 0.1, 0), (1, 10, true_split, 0.1, 1)], [], 2,
                 dict(cpfit=cpfit, true_eps=False, smooth=True, unfolded=False), 0,
                 np.full(n_start, float(true_split)), par, dict(split=true_split))
    w.jsfs = np.array([synth.counts_from_spectrum(jafs, n_sites)])
    return w


def config4(spectrum_fn, n_split=256, n_rep=1000, true_split=50, cpfit=False, seed=3, n_sites=10 ** 6, psmc_seeds=(1, 2), truth_seed=7):
    """No migration; split scan incl. fractional values x bootstrap replicates
    (test.bs/*no.mig.sh shape; replicate table as utils/generateJSFS_bs.py writes it)."""
    inp = synth.psmc_pair(64, 65, seeds=psmc_seeds)
    times, lh, _ = synth.self_consistent(inp, true_split, seed=truth_seed)
    jafs = spectrum_fn(times, lh, true_split, [], [], 0)
    row = synth.counts_from_spectrum(jafs, n_sites)
    table = mio.bootstrap_table(synth.chunk_rows(row, 20), n_rep - 1, random.Random(seed))
    splits = 20.0 + 0.25 * np.arange(n_split)
    w = Workload("config4: numT=128, no migration, %d split values x %d bootstrap replicates" % (n_split, n_rep),
                 times, lh, [], [], 0, dict(cpfit=cpfit, true_eps=False, smooth=True, unfolded=False), 0,
                 splits, None, dict(split=true_split))
    w.jsfs = np.array(table, dtype=np.float64)
    return w


def config5(spectrum_fn, n_split=32, n_rate=64, n_pulse=32, first_split=64, true_split=80, n_sites=10 ** 6, cpfit=True, psmc_seeds=(1, 2), truth_seed=7):
    """Ancient second genome (--sdate 40000) + --hetloss 0 0.1; split x band rate x pulse fraction."""
    inp = synth.psmc_pair(64, 64, seeds=psmc_seeds, sample_date=40000.0, units=mio.Units(hetloss2=0.1))
    sd = inp.sampleDateDiscr
    band_truth = [(0, sd + 2, true_split, 0.15, -1)]
    pulse_truth = [(1, sd + 10, 0.1, -1)]
    mis, pus = _mis_pus(band_truth, pulse_truth, true_split)
    times, lh, _ = synth.self_consistent(inp, true_split, mis, pus, seed=truth_seed)
    jafs = spectrum_fn(times, lh, true_split, band_truth, pulse_truth, sd)
    splits = np.arange(first_split, first_split + n_split, dtype=np.float64)
    rates = np.logspace(-3, 0, n_rate)
    fr = np.linspace(0.0, 0.5, n_pulse)
    a, b, c = np.meshgrid(splits, rates, fr, indexing="ij")
    w = Workload("config5: numT=128 sdate=40000 hetloss=(0,0.1), %dx%dx%d split x mi x pu grid, %s" % (n_split, n_rate, n_pulse, "--cpfit" if cpfit else "default fit"),
                 times, lh, [(0, sd + 2, -1, 0.0, 0)], [(1, sd + 10, 0.0, 1)], 2,
                 dict(cpfit=cpfit, true_eps=False, smooth=True, unfolded=False), sd,
                 a.ravel().copy(), np.stack([b.ravel(), c.ravel()], axis=1), dict(split=true_split, sample_date=sd))
    w.jsfs = np.array([synth.counts_from_spectrum(jafs, n_sites)])
    return w


BUILDERS = {"config1": config1, "config2": config2, "config2x16": config2x16, "config2x2": lambda f: config2x16(f, n_grid=2),
            "config2x4": lambda f: config2x16(f, n_grid=4), "config2x8": lambda f: config2x16(f, n_grid=8), "config3": config3, "config4": config4, "config5": config5}


def dump_text(workload, path, spectrum_fn=None, **kw):
    """Write a workload as the whitespace-separated text file examples/lanes_throughput.c reads (format: its header comment).  `workload` is a
    Workload or the name of one of the constructors above (then the truth spectrum comes from the engine on device 0 unless spectrum_fn is given)."""
    if isinstance(workload, str):
        if spectrum_fn is None:
            from .engine import truth_spectrum
            spectrum_fn = lambda *a: truth_spectrum(*a)
        workload = globals()[workload](spectrum_fn, **kw)
    w = workload
    g = lambda v: "%.17g" % float(v)
    flags = (1 if w.flags.get("cpfit") else 0) | (2 if w.flags.get("true_eps") else 0) | (4 if w.flags.get("smooth") else 0) | (8 if w.flags.get("unfolded") else 0)
    rows = np.atleast_2d(np.asarray(w.jsfs, dtype=np.float64))
    out = ["%d %d %d %d %d %d %s" % (w.numT, w.sample_date, flags, len(w.bands), len(w.pulses), w.n_param, g(0.0))]
    out.append(" ".join(g(t) for t in w.times))
    out.append(" ".join(g(v) for pair in w.lh for v in pair))
    for pop, start, end, value, param in w.bands:
        out.append("%d %d %d %d %s" % (pop, start, end, param, g(value)))
    for pop, time, value, param in w.pulses:
        out.append("%d %d %d %s" % (pop, time, param, g(value)))
    out.append("%d %d" % (w.n_cand, rows.shape[0]))
    out.append(" ".join(g(s) for s in w.split_time))
    if w.n_param:
        out.append(" ".join(g(v) for v in np.asarray(w.params, dtype=np.float64).reshape(-1)))
    out.append(" ".join(g(v) for v in rows.reshape(-1)))
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    return path
