"""Host-side readers/writers for the text formats either side of the hot path.

These mirror the behaviour of the reference's ``migrationIO.py`` for the file
formats MiSTI consumes and produces (cites are ``/root/reference`` file:line):

  * PSMC output  -> merged two-genome time grid   (ReadPSMCFile :183-222, ReadPSMC :224-295)
  * units file   -> scaling constants              (Units :100-176)
  * JSFS file    -> rows of 8 numbers              (ReadJAFS :557-608, PrintJAFSFile :526-554)
  * bootstrap resampling of JSFS chunks            (BootstrapJAFS :506-524)
  * ``#MiSTI2 ver 0.4`` result file                (OutputMigration :346-375)

Everything here is microsecond-scale text handling on the host; none of it is
on the likelihood hot path.
"""
from __future__ import annotations

import io as _io
import random
from dataclasses import dataclass, field

import numpy as np

JSFS_COLUMNS = ["total", "0100", "1100", "0001", "0101", "1101", "0011", "0111"]


class FormatError(ValueError):
    """A malformed input file (the reference prints and calls sys.exit(0))."""


@dataclass
class Units:
    """Scaling constants; defaults of migrationIO.Units (:101-107).

    Unlike the reference (class-level mutable state) this is an ordinary value.
    """
    mutRate: float = 1.25e-8
    binsize: float = 100
    N0: float = 10000
    genTime: float = 1
    hetloss1: float = 0.0
    hetloss2: float = 0.0

    @classmethod
    def from_file(cls, fn, **kw):
        """``key=value`` lines, unknown keys ignored, missing file -> defaults (:146-176)."""
        u = cls(**kw)
        try:
            with open(fn) as f:
                for line in f:
                    parts = line.split("=")
                    if len(parts) != 2 or parts[0] not in ("mutRate", "binsize", "N0", "genTime"):
                        continue
                    try:
                        setattr(u, parts[0], float(parts[1]))
                    except ValueError:
                        print("Cannot read %s entry from file, using default or previous values" % parts[0])
        except OSError:
            print("Units input file not found, using default values.")
        return u

    def set_hetloss(self, hl):                                   # SetHetLoss :129-141
        for name, v in zip(("hetloss1", "hetloss2"), hl):
            if v is None:
                continue
            if not (0.0 <= v < 1.0):
                raise FormatError("Hetloss should be between 0 and 1.")
            setattr(self, name, float(v))

    def describe(self):                                          # PrintUnits :143-144
        return ("Units: mutation rate = %s \tbinsize = %s \tN0 = %s \tgeneration time = %s"
                % (self.mutRate, self.binsize, self.N0, self.genTime))


@dataclass
class InputData:
    """What ReadPSMC returns (migrationIO.InputData :46-58)."""
    times: list            # numT-1 interval lengths, coalescent units
    lambdas: list          # numT pairs of coalescence rates
    scaleTime: float
    theta: float
    divergenceTime: float = -1
    scaleEPS: float = 1.0
    rho: float | None = None
    sampleDateDiscr: int = 0
    Tpsmc: list = field(default_factory=list)
    mi: list | None = None         # [[pop(1|2), start, end, rate, optimise(0|1)], ...]   (read_ms)
    pu: list | None = None         # [[pop(1|2), interval, fraction, optimise(0|1)], ...]


def read_psmc_file(src, rd=-1):
    """One PSMC output -> (times, sizes, round, theta, rho).  ReadPSMCFile :183-222.

    ``src`` is a path or a file-like object.  The last ``RD`` round is used when
    ``rd`` is -1 or exceeds the number of rounds.
    """
    if hasattr(src, "read"):
        lines = src.read().splitlines()
    else:
        with open(src) as f:
            lines = f.read().splitlines()
    rows = [l.split() for l in lines]
    if any(len(r) == 0 for r in rows):
        raise FormatError("blank line in PSMC file")          # reference: IndexError
    rounds = [int(r[1]) for r in rows if r[0] == "RD"]
    if not rounds:
        raise FormatError("Corrupted or empty input file")
    last = rounds[-1]                                            # reference keeps the last seen
    if rd == -1 or rd > last:
        rd = last
    start = next(i for i, r in enumerate(rows) if r[0] == "RD" and int(r[1]) == rd)
    th = rh = 0.0
    i = start
    while rows[i][0] != "RS":
        if rows[i][0] == "TR":
            th, rh = float(rows[i][1]), float(rows[i][2])
        i += 1
        if i >= len(rows):
            raise FormatError("no RS rows after RD %d" % rd)
    tk, lk = [], []
    while rows[i][0] != "PA":
        if rows[i][0] != "RS":
            raise FormatError("Unexpected line.")
        tk.append(float(rows[i][2]))
        lk.append(float(rows[i][3]))
        i += 1
        if i >= len(rows):
            raise FormatError("PSMC round not terminated by a PA line")
    return tk, lk, rd, th, rh


def merge_psmc(d1, d2, sample_date=0.0, units=None):
    """Two parsed PSMC trajectories -> InputData on the merged grid.  ReadPSMC :224-295."""
    u = units or Units()
    theta = 4.0 * u.binsize * u.mutRate * u.N0
    scale_time = 2 * u.genTime * u.N0
    th1 = d1[3] / (1.0 - u.hetloss1)
    th2 = d2[3] / (1.0 - u.hetloss2)
    t1 = [v * th1 / theta for v in d1[0]]
    e1 = [v * th1 / theta for v in d1[1]]
    t2 = [v * th2 / theta for v in d2[0]]
    e2 = [v * th2 / theta for v in d2[1]]
    sd = sample_date / 2 / u.N0 / u.genTime
    if sd > 0:                                                   # ancient second genome :244-248
        t2 = [0.0] + [v + sd for v in t2]
        e2 = [1.0] + e2
    grid = sorted(t1 + t2[1:])
    try:
        sample_discr = grid.index(sd)
    except ValueError:
        raise FormatError("sample date is not a point of the merged grid")

    def rates_on_grid(t, e):
        # rate of the genome's interval that contains each merged grid point
        idx = np.searchsorted(np.asarray(t[1:]), np.asarray(grid), side="right")
        idx = np.minimum(idx, len(e) - 1)
        own = [0] + [int(np.searchsorted(np.asarray(grid), v, side="left")) for v in t[1:]] + [len(grid)]
        return [1.0 / e[i] for i in idx], own

    l1, own1 = rates_on_grid(t1, e1)
    l2, own2 = rates_on_grid(t2, e2)
    lam = [[a, b] for a, b in zip(l1, l2)]
    dt = [b - a for a, b in zip(grid[:-1], grid[1:])]
    return InputData(dt, lam, scale_time, theta, scaleEPS=1, rho=d1[4] * theta / th1,
                     sampleDateDiscr=sample_discr, Tpsmc=[own1, own2])


def read_psmc(fn1, fn2, sample_date=0.0, rd=-1, units=None):
    return merge_psmc(read_psmc_file(fn1, rd), read_psmc_file(fn2, rd), sample_date, units)


# ------------------------------------------------------- ms command line ----
# ms options the reader understands: name -> number of values.  Everything else is skipped one
# token at a time, exactly as the reference does (so "-t 15000", "-I 2 2 2", "-eM ..." fall
# through harmlessly).
_MS_ARITY = {"-n": 2, "-en": 3, "-eN": 2, "-em": 4, "-es": 3, "-ej": 3}


def read_ms(argument_string):
    """An ``ms``/``scrm`` command line of a two-population split model -> model inputs
    (``migrationIO.ReadMS``, migrationIO.py:659-753; used by ``TestModel.py:88``).

    Times come out as interval lengths in coalescent units (2 x the ms time differences),
    rates as 1/size, migration bands as ``[pop, first interval, end interval, 2 x ms rate, 0]``
    (a band ends where the next ``-em`` of the same population starts, the last one at the split),
    ``-es t i p`` as a pulse of fraction ``1 - p`` into population ``i``.  Sizes set with ``-eN``
    apply to both populations; from the split on, the merged-away population copies the other one.
    Like the reference, the reader trusts the command line: ``-ej`` must name populations 1/2.
    """
    tok = argument_string.split(" ")
    size_at = [{0.0: 1.0}, {0.0: 1.0}]             # per population: time -> relative size
    band_at = [{}, {}]                             # per receiving population: time -> ms rate
    pulse_at = {}                                  # time -> (fraction, population)
    split_time, moved = 0, None
    i = 0
    while i < len(tok):
        opt = tok[i]
        n = _MS_ARITY.get(opt)
        if n is None:
            i += 1
            continue
        a = tok[i + 1: i + 1 + n]
        if opt in ("-n", "-en"):
            when, pop, size = (0.0, int(a[0]), float(a[1])) if opt == "-n" else (float(a[0]), int(a[1]), float(a[2]))
            if pop not in (1, 2):
                print("Population id should be 1 or 2.")
                print(" ".join(tok[i: i + 1 + n]))
                raise SystemExit(0)
            size_at[pop - 1][when] = size
        elif opt == "-eN":
            size_at[0][float(a[0])] = size_at[1][float(a[0])] = float(a[1])
        elif opt == "-em":
            pop = int(a[1])
            band_at[pop - 1][float(a[0])] = float(a[3])
        elif opt == "-es":
            pulse_at[float(a[0])] = (1 - float(a[2]), int(a[1]))
        elif opt == "-ej" and int(a[1]) <= 2:
            split_time, moved = float(a[0]), int(a[1]) - 1
        i += 1 + n
    if moved is None:
        print("Populations should be merged. (-ej [time] 2 1)")
        raise SystemExit(0)
    knots = sorted(set(size_at[0]) | set(size_at[1]) | set(band_at[0]) | set(band_at[1]) | set(pulse_at) | {split_time})
    index = {t: k for k, t in enumerate(knots)}
    split = index[split_time]
    sizes = [[0, 0] for _ in knots]
    for k in (0, 1):
        cur = 0
        for j, t in enumerate(knots):
            v = size_at[k].get(t, 0)
            cur = v if v != 0 else cur              # a size of exactly 0 means "unchanged", as in the reference
            sizes[j][k] = cur
    for j in range(split, len(knots)):
        sizes[j][moved] = sizes[j][1 - moved]
    mis = []
    for k in (0, 1):
        starts = sorted(band_at[k])
        for a, t in enumerate(starts):
            end = index[starts[a + 1]] if a + 1 < len(starts) else split
            mis.append([k + 1, index[t], end, 2 * band_at[k][t], 0])
    pus = [[pop, index[t], frac, 0] for t, (frac, pop) in pulse_at.items()]
    times = [2 * (b - a) for a, b in zip(knots[:-1], knots[1:])]
    lambdas = [[1.0 / u, 1.0 / v] for u, v in sizes]
    return InputData(times, lambdas, 1.0, 1.0, divergenceTime=split, mi=mis, pu=pus)


# ---------------------------------------------------------------- JSFS ----
def read_jsfs(src):
    """JSFS file (format version >= 1) -> (rows, pop1, pop2).  ReadJAFS :557-608."""
    if hasattr(src, "read"):
        lines = src.read().split("\n")
    else:
        with open(src) as f:
            lines = f.read().split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    if not lines or not lines[0].startswith("#MiSTI_JSFS"):
        raise FormatError("Corrupted JSFS file header.")
    pars = lines[0].split(" ")
    if len(pars) < 3 or float(pars[2]) < 1:
        raise FormatError("The file version is not supported anymore.")
    pop1 = pop2 = None
    i = 0
    while i < len(lines) and lines[i].startswith("#"):
        tag = lines[i][1:5]
        if tag in ("pop1", "pop2"):
            p = lines[i].split("\t")
            if len(p) != 2:
                raise FormatError("Corrupted JSFS file header.")
            if tag == "pop1":
                pop1 = p[1]
            else:
                pop2 = p[1]
        i += 1
    if i < len(lines) and lines[i].startswith("total"):
        i += 1
    rows = []
    for line in lines[i:]:
        cols = line.split("\t")
        if len(cols) != 8:
            raise FormatError("Unexpected line. Expected an entry for JSFS with eight TAB-separated columns.")
        rows.append([float(v) for v in cols])
    return rows, pop1, pop2


def format_jsfs(rows, pop1=None, pop2=None):
    """Rows (7 or 8 numbers each) -> JSFS file text.  PrintJAFSFile :526-554."""
    out = ["#MiSTI_JSFS version 1.0"]
    if pop1:
        out.append("#pop1\t" + pop1.strip("\n\r"))
    if pop2:
        out.append("#pop2\t" + pop2.strip("\n\r"))
    out.append("\t".join(JSFS_COLUMNS))
    if rows and not isinstance(rows[0], (list, tuple, np.ndarray)):
        rows = [rows]
    for r in rows:
        r = list(r)
        if len(r) == 7:
            r = [sum(r)] + r
        if len(r) != 8:
            raise FormatError("Unexpected SFS entry.")
        out.append("\t".join(str(v) for v in r))
    return "\n".join(out) + "\n"


def bootstrap_jsfs(rows, rng=None, normalize=False):
    """Resample chunk rows with replacement up to the genome length.  BootstrapJAFS :506-524."""
    rng = rng or random
    if any(len(r) != 8 for r in rows):
        raise FormatError("Cannot use provided SFS for bootstrap.")
    genome = sum(r[0] for r in rows)
    seg = sum(sum(r[1:]) for r in rows)
    sfs = [0] * 8
    while sfs[0] < genome:
        pick = rows[rng.randint(0, len(rows) - 1)]
        sfs = [a + b for a, b in zip(sfs, pick)]
    if normalize:
        seg_bs = sum(sfs[1:])
        sfs = [v * (seg / seg_bs) for v in sfs]
    return sfs


def bootstrap_table(rows, n, rng=None):
    """Row 0 = sum of all chunks, rows 1..n = resamples (utils/generateJSFS_bs.py:39-48)."""
    true = [sum(r[i] for r in rows) for i in range(8)]
    return [true] + [bootstrap_jsfs(rows, rng) for _ in range(n)]


# ------------------------------------------------------ result writer ----
def format_migration(model, llh, scale_time=1, scale_eps=1):
    """``#MiSTI2 ver 0.4`` text from an evaluated engine.  OutputMigration :346-369.

    ``model`` needs ``times splitT sampleDate thrh JAFS dataJAFS lc lh mi Pr``.
    """
    acc = [sum(model.times[0:i]) for i in range(len(model.times) + 1)]
    o = _io.StringIO()
    o.write("#MiSTI2 ver 0.4\n")
    o.write("LK\t%s\n" % llh)
    o.write("ST\t%s\n" % model.splitT)
    o.write("SD\t%s\n" % model.sampleDate)
    o.write("TR\t%s\t%s\n" % (model.thrh[0], model.thrh[1]))
    o.write("SFS\t" + "\t".join(map(str, model.JAFS)) + "\n")
    tot = sum(model.dataJAFS)
    o.write("DSF\t" + "\t".join(str(v / tot) for v in model.dataJAFS) + "\n")
    o.write("SCT\t%s\n" % scale_time)
    o.write("SCE\t%s\n" % scale_eps)
    for i, t in enumerate(acc):
        cols = [t, 1.0 / model.lc[i][0], 1.0 / model.lc[i][1], 1.0 / model.lh[i][0], 1.0 / model.lh[i][1],
                model.mi[i][0], model.mi[i][1]]
        if i < model.splitT:
            for pair in model.Pr[i]:
                cols += [pair[0], pair[1]]
        o.write("RS\t" + "\t".join(str(v) for v in cols) + "\n")
    return o.getvalue()


# ------------------------------------------------------ result reader ----
class MigData:
    """What ``ReadMigration`` returns (migrationIO.MigData, /root/reference/migrationIO.py:65-99): attribute names as there."""

    def __init__(self):
        self.llh = self.splitT = self.migStart = self.migEnd = self.times = None
        self.lambda1 = self.lambda2 = self.lambdah1 = self.lambdah2 = self.thrh = self.mi = self.sampleDate = None
        self.jaf = None
        self.pr = None                 # [(p11_1, p11_2, p22_1, p22_2, p12_1, p12_2)] per RS row (zeros where the row has none)


def read_migration(path_or_file, scale_time=1, scale_eps=1):
    """Reader of the ``-o`` result file (``#MiSTI2 ver 0.3 / 0.4``; ``ReadMigration``, /root/reference/migrationIO.py:377-504,
    without its plotting): times are scaled by the file's ``SCT`` (default ``scale_time``), rates are ``1 / N`` divided by
    its ``SCE``.  ``jaf`` is a list (the reference leaves a ``map`` object)."""
    fh = open(path_or_file) if isinstance(path_or_file, str) else path_or_file
    d = MigData()
    times, lc1, lc2, lh1, lh2, pr = [], [], [], [], [], []
    try:
        head = next(fh).rstrip().split(" ")
        if head[0] != "#MiSTI2":
            raise ValueError("not a #MiSTI2 result file (the pre-0.3 format is not supported)")
        version = float(head[2])
        if version < 0.3:
            raise ValueError("File version is not supported anymore.")
        for line in fh:
            f = line.rstrip("\n").split("\t")
            tag = f[0]
            if tag == "LK":
                d.llh = float(f[1])
            elif tag == "ST":
                d.splitT = int(f[1])
            elif tag == "SD":
                d.sampleDate = int(f[1])
            elif tag == "TR":
                d.thrh = [float(f[1]), float(f[2])]
            elif tag == "SFS":
                d.jaf = [float(v) for v in f[1:]]
            elif tag == "SCT":
                scale_time = float(f[1])
            elif tag == "SCE":
                scale_eps = float(f[1])
            elif tag == "RS":
                times.append(float(f[1]) * scale_time)
                lc1.append(1.0 / float(f[2]) / scale_eps)
                lc2.append(1.0 / float(f[3]) / scale_eps)
                shift = 0
                if version >= 0.4:
                    lh1.append(1.0 / float(f[4]) / scale_eps)
                    lh2.append(1.0 / float(f[5]) / scale_eps)
                    shift = 2
                pr.append(tuple(float(v) for v in f[6 + shift:12 + shift]) if len(f) > 6 + shift else (0,) * 6)
    finally:
        if isinstance(path_or_file, str):
            fh.close()
    d.times, d.lambda1, d.lambda2, d.lambdah1, d.lambdah2, d.pr = times, lc1, lc2, lh1, lh2, pr
    return d
