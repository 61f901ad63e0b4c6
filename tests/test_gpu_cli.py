"""The MiSTI.py-compatible command line end to end on synthetic files (config 1 of
BASELINE.json: numT = 32, split 20, no migration, single likelihood) and an optimised band."""
import io
import os
import re
import contextlib

import numpy as np
import pytest

from parity import llk_tol

pytestmark = pytest.mark.gpu


def write_inputs(tmp_path, n1=16, n2=17):
    from misti_amd import synth, io as mio
    from oracle.batch import oracle_truth_spectrum
    f1, f2, fj = (str(tmp_path / n) for n in ("g1.psmc", "g2.psmc", "data.sfs"))
    open(f1, "w").write(synth.psmc_text(n1, 1, synth.THETA_1))
    open(f2, "w").write(synth.psmc_text(n2, 2, synth.THETA_2))
    inp = mio.read_psmc(f1, f2)
    jafs = oracle_truth_spectrum(inp.times, inp.lambdas, 20, [], [], 0)
    row = synth.counts_from_spectrum(jafs, 200000)
    open(fj, "w").write(mio.format_jsfs(synth.chunk_rows(row, 5)))
    return f1, f2, fj, inp, row


def run_cli(args):
    from misti_amd import cli
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        rc = cli.main(args)
    return rc, out.getvalue()


def test_single_likelihood_matches_oracle(tmp_path):
    from oracle.misti_oracle import OracleModel
    f1, f2, fj, inp, row = write_inputs(tmp_path)
    units = str(tmp_path / "nounits.txt")
    rc, text = run_cli([f1, f2, fj, "20", "--funits", units])
    assert rc == 0
    m = re.search(r"bs_id = -1 \tsplitT = 20.0 \ttime = (\S+) \tmigration rates  \tllh = (\S+)", text)
    assert m, text
    llh = float(m.group(2))
    o = OracleModel(inp.times, inp.lambdas, row, 20, [], [], smooth=True)
    want = o.jafs_likelihood([])
    assert abs(llh - want) <= llk_tol(want, row, o.JAFS, False)
    assert float(m.group(1)) == pytest.approx(sum(inp.times[:20]) * inp.scaleTime, rel=1e-14)
    assert "Total number of likelihood function calls is" in text


def test_output_file_and_fractional_split(tmp_path):
    f1, f2, fj, inp, row = write_inputs(tmp_path)
    out = str(tmp_path / "res.mi")
    rc, text = run_cli([f1, f2, fj, "19.5", "--cpfit", "-bs", "0", "-o", out, "--funits", str(tmp_path / "x")])
    assert rc == 0 and os.path.exists(out)
    lines = open(out).read().splitlines()
    assert lines[0] == "#MiSTI2 ver 0.4" and lines[1].startswith("LK\t") and lines[2] == "ST\t20"
    rs = [l for l in lines if l.startswith("RS\t")]
    assert len(rs) == 33                              # numT + 1 rows after the inserted interval
    assert len(rs[0].split("\t")) == 14 and len(rs[25].split("\t")) == 8   # Pr columns only before the split


def test_optimised_band_and_grid(tmp_path):
    f1, f2, fj, inp, row = write_inputs(tmp_path)
    rc, text = run_cli([f1, f2, fj, "20", "-mi", "1", "2", "20", "0.1", "1", "--cpfit", "-tol", "1e-3", "--funits", str(tmp_path / "x")])
    assert rc == 0
    m = re.search(r"optim = \[(\S+)\] \tllh = (\S+)", text)
    assert m, text[-600:]
    assert float(m.group(1)) >= 0 and np.isfinite(float(m.group(2)))
    rc, text = run_cli([f1, f2, fj, "20", "-mi", "1", "2", "20", "0.1", "1", "--cpfit", "--grid-st", "18", "22",
                        "--grid-mi", "0", "0.001", "0.1", "4", "--all-bs", "--funits", str(tmp_path / "x")])
    assert rc == 0
    assert len(re.findall(r"^bs_id = ", text, flags=re.M)) == 5 * 4 * 5
    assert "best: splitT =" in text
    m = re.search(r"bootstrap: best splitT per replicate mean = (\S+), 95% interval = \[(\S+), (\S+)\] over 5 replicates", text)
    assert m, text[-800:]
    mean, lo, hi = (float(v) for v in m.groups())
    assert 18 <= lo <= mean <= hi <= 22
    # the same sweep on a device LIST from this one process (misti_create_multi; two contexts on the one GPU of the box): the same lines
    rc2, text2 = run_cli([f1, f2, fj, "20", "-mi", "1", "2", "20", "0.1", "1", "--cpfit", "--grid-st", "18", "22",
                          "--grid-mi", "0", "0.001", "0.1", "4", "--all-bs", "--funits", str(tmp_path / "x"), "--devices", "0,0"])
    rows = lambda t: [l for l in t.splitlines() if l.startswith("bs_id =") or l.startswith("best:") or l.startswith("bootstrap:")]
    assert rc2 == 0 and rows(text2) == rows(text)
    # ... and with one rank per GPU started by the command itself (--gpus 1 here is the plain path; N > 1 over gloo: tests/test_dist_cpu.py)
    rc3, text3 = run_cli([f1, f2, fj, "20", "-mi", "1", "2", "20", "0.1", "1", "--cpfit", "--grid-st", "18", "22",
                          "--grid-mi", "0", "0.001", "0.1", "4", "--all-bs", "--funits", str(tmp_path / "x"), "--gpus", "1"])
    assert rc3 == 0 and rows(text3) == rows(text)
    # ... and as a rank of a (one-rank) RCCL process group, started the way --gpus N starts its ranks: the group is initialised on the GPU
    # (backend nccl = RCCL), the sweep runs, the group is left in order
    import subprocess
    import sys
    from conftest import ROOT
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29541",
           "-m", "misti_amd.cli", f1, f2, fj, "20", "-mi", "1", "2", "20", "0.1", "1", "--cpfit", "--grid-st", "18", "22",
           "--grid-mi", "0", "0.001", "0.1", "4", "--all-bs", "--funits", str(tmp_path / "x")]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-1500:]
    assert rows(r.stdout) == rows(text)


def _tokens_match(ours, theirs, rtol):
    """Same text structure; numeric tokens agree to rtol, everything else byte for byte."""
    a, b = ours.replace("\t", " \t ").split(" "), theirs.replace("\t", " \t ").split(" ")
    assert len(a) == len(b), (ours, theirs)
    for x, y in zip(a, b):
        try:
            fy = float(y)
        except ValueError:
            assert x == y, (ours, theirs)
            continue
        fx = float(x)
        assert fx == pytest.approx(fy, rel=rtol, abs=1e-300), (x, y, ours, theirs)


def _reference_spread(fx, run, kinds=8):
    """Largest relative change of the oracle's llh for this MiSTI.py run under 2^-48 perturbations of its inputs."""
    import argparse
    import warnings
    from misti_amd import io as mio
    from oracle.batch import oracle_eval
    from parity import perturbed
    inp = mio.merge_psmc(mio.read_psmc_file(io.StringIO(fx["psmc1"])), mio.read_psmc_file(io.StringIO(fx["psmc2"])))
    rows, _, _ = mio.read_jsfs(io.StringIO(fx["jsfs"]))
    ap = argparse.ArgumentParser()
    ap.add_argument("split", type=float)
    ap.add_argument("-mi", nargs=5, action="append", default=[])
    ap.add_argument("--cpfit", action="store_true")
    ap.add_argument("-uf", action="store_true")
    ap.add_argument("-bs", type=int, default=-1)
    ap.add_argument("-o", default="")
    a = ap.parse_args(run["args"])
    row = rows[a.bs] if a.bs >= 0 else [sum(r[i] for r in rows) for i in range(8)]
    bands = [(int(m[0]) - 1, int(m[1]), int(m[2]), float(m[3]), -1) for m in a.mi]
    flags = dict(cpfit=a.cpfit, true_eps=False, smooth=True, unfolded=a.uf)
    times, lh = list(inp.times), [list(x) for x in inp.lambdas]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        base = oracle_eval(times, lh, bands, [], flags, 0, a.split, [], [row])[0][0]
        vals = [oracle_eval(*perturbed(times, lh, k), bands, [], flags, 0, a.split, [], [row])[0][0] for k in range(kinds)]
    return max(abs(v - base) / abs(base) for v in vals if np.isfinite(v))


def test_cli_against_reference_runs(tmp_path):
    """MiSTI.py itself was run on these files (tests/golden/make_golden.py, one OS process per run): the machine-read
    result line (MiSTI.py:240, what the test.bs scripts grep) and the -o file (migrationIO.OutputMigration :346-375)
    must come out the same - structure byte for byte, numbers to the likelihood tolerance."""
    import json
    from conftest import GOLDEN
    fx = json.load(open(os.path.join(GOLDEN, "golden_host.json")))["cli"]
    for name, key in (("g1.psmc", "psmc1"), ("g2.psmc", "psmc2"), ("data.sfs", "jsfs"), ("setunits.txt", "units")):
        (tmp_path / name).write_text(fx[key])
    for run in fx["runs"]:
        res = tmp_path / "res.mi"
        if res.exists():
            res.unlink()
        from misti_amd.engine import MigrationInference as MIc
        MIc.COUNT_LLH = MIc.CORRECTION_CALLED = MIc.CORRECTION_FAILED = 0     # per-process counters in the reference (one process per run there)
        rc, text = run_cli(["g1.psmc", "g2.psmc", "data.sfs"] + run["args"] + ["-wd", str(tmp_path), "--funits", str(tmp_path / "setunits.txt")])
        assert rc == 0
        line = [l for l in text.splitlines() if l.startswith("bs_id =")]
        assert len(line) == 1, text[-800:]
        # The per-candidate contract (tests/parity.py): 1e-8 on the printed value, or 10 x the reference's own indeterminacy for
        # THIS run - measured here with the oracle (which reproduces these four reference runs bit for bit) under eight 2^-48
        # perturbations of the inputs.  The fourth run (a one-way band, --cpfit) ends in a runaway solve that walks 35 steps to
        # rate x length 86 478 and stops on a gradient test whose noise is 20 % of gtol: the reference itself moves by 3.9e-6.
        rtol = 1e-8
        try:
            _tokens_match(line[0], run["result_line"], rtol)
        except AssertionError:
            rtol = 10.0 * _reference_spread(fx, run)
            assert rtol > 1e-8, "the reference determines this run, the HIP path misses it"
            _tokens_match(line[0], run["result_line"], rtol)
        # the lines around it that scripts may rely on
        for must in ("Reading from files:", "Parameter estimates:", "Total number of likelihood function calls is 1",
                     "Lambda correction called 1 times.", "Lambda correction failed 0 times."):
            assert must in text and must in run["stdout"], must
        if run["out_file"] is None:
            assert not res.exists()                       # -o is honoured with -bs 0 only (MiSTI.py:248-249)
            continue
        ours, theirs = res.read_text().splitlines(), run["out_file"].splitlines()
        assert len(ours) == len(theirs)
        for a, b in zip(ours, theirs):
            assert a.split("\t")[0] == b.split("\t")[0]
            if rtol > 1e-8 and a.startswith("RS"):
                assert len(a.split("\t")) == len(b.split("\t"))          # a runaway rate is in these rows: structure only
                continue
            _tokens_match(a, b, 1e-6 if a.startswith("RS") else max(rtol, 1e-8))


def test_result_writer_on_the_engine_matches_the_reference_text():
    """OutputMigration for the anchors A3 (optimised bands, unfolded), A6 (ancient sample) and A7 (fractional split):
    io.format_migration on the GPU-backed mirror against the reference's own text."""
    import json
    from conftest import GOLDEN
    from misti_amd import io as mio
    from misti_amd.engine import MigrationInference
    for c in json.load(open(os.path.join(GOLDEN, "golden_host.json")))["writer"]:
        i = c["in"]
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            m = MigrationInference(list(i["times"]), [list(x) for x in i["lambdas"]], list(i["sfs"]), i["split"], [list(x) for x in i["mi"]],
                                   [list(x) for x in i["pu"]], thrh=i["thrh"], **i["kw"])
            llh = m.JAFSLikelihood(list(i["params"]))
        text = mio.format_migration(m, llh, i["scaleTime"], i["scaleEPS"])
        ours, theirs = text.splitlines(), c["text"].splitlines()
        theirs = [l for l in theirs if l != ""]
        assert len(ours) == len(theirs), c["name"]
        for a, b in zip(ours, theirs):
            _tokens_match(a, b, 1e-6 if a.startswith("RS") else 1e-9)
