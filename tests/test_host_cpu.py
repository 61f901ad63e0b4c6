"""CPU-side tests: readers against the reference's own reader dumps, C-ABI library
loads and exports every declared symbol, the product path fails loudly without a GPU."""
import ctypes as C
import io
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from misti_amd import _lib, io as mio, synth


def reader_cases():
    return json.load(open(os.path.join(GOLDEN, "golden_readers.json")))["cases"]


@pytest.mark.parametrize("k", range(4))
def test_read_psmc_matches_reference_dump(k):
    c = reader_cases()[k]
    hl = c["hetloss"] or (0.0, 0.0)
    u = mio.Units(hetloss1=hl[0], hetloss2=hl[1])
    d = mio.merge_psmc(mio.read_psmc_file(io.StringIO(c["psmc1"])), mio.read_psmc_file(io.StringIO(c["psmc2"])), c["sdate"], u)
    assert d.times == c["times"]
    assert d.lambdas == c["lambdas"]
    assert d.sampleDateDiscr == c["sampleDateDiscr"]
    assert d.Tpsmc == c["Tpsmc"]
    assert d.theta == c["theta"] and d.rho == c["rho"] and d.scaleTime == c["scaleTime"]


def test_psmc_round_selection_and_errors():
    t = synth.psmc_text(8, 3, 0.05, rounds=3)
    a = mio.read_psmc_file(io.StringIO(t), rd=-1)
    b = mio.read_psmc_file(io.StringIO(t), rd=99)
    c = mio.read_psmc_file(io.StringIO(t), rd=0)
    assert a[2] == 2 and b[2] == 2 and c[2] == 0 and a[:2] != c[:2]
    with pytest.raises(mio.FormatError):
        mio.read_psmc_file(io.StringIO("XX 1\n"))


def test_jsfs_roundtrip_and_bootstrap():
    rows = [[1000.0, 10, 20, 30, 40, 50, 60, 70], [900.0, 1, 2, 3, 4, 5, 6, 7]]
    text = mio.format_jsfs(rows, "popA", "popB")
    back, p1, p2 = mio.read_jsfs(io.StringIO(text))
    assert back == rows and p1 == "popA" and p2 == "popB"
    assert text.splitlines()[0] == "#MiSTI_JSFS version 1.0"
    assert mio.format_jsfs([[1, 2, 3, 4, 5, 6, 7]]).splitlines()[-1].split("\t")[0] == "28"
    import random
    tab = mio.bootstrap_table(rows, 5, random.Random(1))
    assert len(tab) == 6 and tab[0] == [1900.0, 11, 22, 33, 44, 55, 66, 77]
    for r in tab[1:]:
        assert r[0] >= 1900.0                       # resampled up to the genome length (BootstrapJAFS :516)
    with pytest.raises(mio.FormatError):
        mio.read_jsfs(io.StringIO("#wrong header\n"))
    with pytest.raises(mio.FormatError):
        mio.read_jsfs(io.StringIO("#MiSTI_JSFS version 1.0\ntotal\n1\t2\n"))


def host_fixtures():
    return json.load(open(os.path.join(GOLDEN, "golden_host.json")))


def test_result_writer_reproduces_the_reference_text():
    """io.format_migration on the reference's own attribute values gives the reference's `#MiSTI2 ver 0.4` text
    byte for byte (migrationIO.OutputMigration :346-375; A3 optimised bands + unfolded, A6 ancient sample, A7
    fractional split)."""
    import types
    for c in host_fixtures()["writer"]:
        m = types.SimpleNamespace(**{k: c["model"][k] for k in ("times", "splitT", "sampleDate", "thrh", "JAFS", "dataJAFS", "lc", "lh", "mi", "Pr")})
        text = mio.format_migration(m, c["model"]["llh"], c["in"]["scaleTime"], c["in"]["scaleEPS"])
        assert text + "\n" == c["text"], c["name"]            # the reference prints the block (print adds the final newline)


def test_bootstrap_resampling_reproduces_the_reference_sequence():
    """BootstrapJAFS under random.seed(k) (migrationIO.py:506-524) and the table of utils/generateJSFS_bs.py:39-48."""
    import random
    b = host_fixtures()["bootstrap"]
    for d in b["draws"]:
        rng = random.Random(d["seed"])
        got = [mio.bootstrap_jsfs(b["rows"], rng, d["normalize"]) for _ in range(3)]
        assert got == d["draws"], (d["seed"], d["normalize"])
    assert mio.bootstrap_table(b["rows"], 4, random.Random(5)) == b["table_seed5"]


def test_units_file(tmp_path):
    f = tmp_path / "u.txt"
    f.write_text("mutRate=2.5e-8\nbinsize=100\nN0=5000\ngenTime=29\njunk\n")
    u = mio.Units.from_file(str(f))
    assert (u.mutRate, u.binsize, u.N0, u.genTime) == (2.5e-8, 100.0, 5000.0, 29.0)
    assert mio.Units.from_file(str(tmp_path / "missing")).N0 == 10000


def test_library_exports_every_declared_symbol():
    """Every function declared in include/misti_hip.h resolves in the built .so."""
    hdr = open(os.path.join(ROOT, "include", "misti_hip.h")).read()
    declared = set(re.findall(r"\b(misti_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.misti_abi_version() == 6 == _lib.ABI_VERSION
    assert "#define MISTI_ABI_VERSION 6" in hdr


def test_tables_match_oracle_structure():
    """The library's own derivation of the 44-state chain equals the oracle's."""
    from oracle.misti_oracle import TWO_POP
    lib = _lib.load()
    gen = np.zeros((4, 44, 44), np.int32)
    jaf = np.zeros((44, 7), np.int32)
    assert lib.misti_tables(gen.ctypes.data_as(C.POINTER(C.c_int32)), jaf.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    assert (gen[0] == TWO_POP.A[0]).all() and (gen[1] == TWO_POP.A[1]).all()
    assert (gen[2] == TWO_POP.B[0]).all() and (gen[3] == TWO_POP.B[1]).all()
    assert (jaf == TWO_POP.jaf).all()
    assert [int((g != 0).sum()) for g in gen] == [44, 44, 88, 88]


def test_no_cpu_fallback():
    """Without a HIP device the product path must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from misti_amd.engine import Engine, MigrationInference
    with pytest.raises(_lib.MistiError) as e:
        Engine([0.1, 0.2], [[1, 1], [1, 1], [1, 1]])
    assert e.value.code == -3
    with pytest.raises(_lib.MistiError):
        MigrationInference([0.1, 0.2], [[1, 1], [1, 1], [1, 1]], [10, 1, 1, 1, 1, 1, 1, 1], 1)


def test_model_validation_messages():
    """Argument errors come back as codes + messages, never exit()."""
    lib = _lib.load()
    times = (C.c_double * 2)(0.1, 0.2)
    lh = (C.c_double * 6)(1, 1, 1, 1, 1, 1)
    band = (_lib.Band * 1)(_lib.Band(0, 2, 1, -1, 0.5))
    m = _lib.Model(3, 0, 0, 1, 0, 0, 0.0, times, lh, band, None)
    ctx = C.c_void_p()
    rc = lib.misti_create(C.byref(m), 0, C.byref(ctx))
    assert rc == -1 and b"strictly less" in lib.misti_last_error()
    m2 = _lib.Model(1, 0, 0, 0, 0, 0, 0.0, times, lh, None, None)
    assert lib.misti_create(C.byref(m2), 0, C.byref(ctx)) == -1


def test_product_never_imports_oracle():
    """No module of the product package may reference the oracle."""
    pkg = os.path.join(ROOT, "misti_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                text = open(os.path.join(base, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "oracle/" not in text, f


def _build_c_example(tmp_path, name="anchor_a2", hip_runtime=False):
    import subprocess
    exe = str(tmp_path / name)
    libdir = os.path.dirname(_lib.lib_path())
    _lib.load()                                              # builds the library if it is missing
    cmd = ["gcc", "-Wall", "-Wextra", "-Werror", "-std=c99", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", name + ".c"), "-L", libdir, "-lmisti_hip", "-Wl,-rpath," + libdir, "-o", exe]
    if hip_runtime:                                          # an example that holds device memory itself: the HIP runtime's C entry points
        cmd += ["-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_header_is_plain_c_and_the_example_links(tmp_path):
    """include/misti_hip.h compiles as C99 (no C++/torch types at the boundary) and a C program links against the .so;
    without a HIP device it reports the fact and exits 2 (no CPU path)."""
    import subprocess
    exe = _build_c_example(tmp_path)
    if _lib.load().misti_device_count() > 0:
        pytest.skip("a GPU is present: the run itself is tests/test_gpu_golden.py::test_c_example")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stderr
    # ... and the multi-device form (misti_create_multi / misti_multi_eval_batch): same header, same behaviour without a device
    exe = _build_c_example(tmp_path, "multi_device")
    r = subprocess.run([exe, "0", "0"], capture_output=True, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stderr
    # ... and the device-resident form with the RCCL gather inside the library (ABI 5): links against the HIP runtime for its own buffers
    exe = _build_c_example(tmp_path, "multi_device_gather", hip_runtime=True)
    r = subprocess.run([exe, "0"], capture_output=True, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stderr
    # ... and the lanes (ABI 6: misti_create_lanes / misti_lanes_eval_batch_dev / misti_lanes_sync)
    exe = _build_c_example(tmp_path, "lanes_throughput", hip_runtime=True)
    r = subprocess.run([exe, os.devnull], capture_output=True, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stderr


def test_engine_context_is_not_destroyed_from_another_process():
    """A forked child (a caller's worker pool) may garbage-collect its copy of an Engine: that copy must not call into
    the library - HIP is unusable in the child (this segfaulted a pool worker and hung the GPU test run once)."""
    import ctypes as C
    import os
    from misti_amd.engine import Engine

    class FakeLib:
        def __init__(self):
            self.destroyed = 0

        def misti_destroy(self, ctx):
            self.destroyed += 1

    e = Engine.__new__(Engine)
    e._lib = FakeLib()
    e._ctx = C.c_void_p(1234)
    e._pid = os.getpid() + 1                  # "created in another process"
    e.close()
    assert e._lib.destroyed == 0 and not e._ctx
    e._ctx = C.c_void_p(1234)
    e._pid = os.getpid()
    e.close()
    assert e._lib.destroyed == 1 and not e._ctx


def test_read_migration_against_the_reference_reader(tmp_path):
    """io.read_migration on the reference's own OutputMigration text against what migrationIO.ReadMigration (:377-504) made
    of the same file (tests/golden/golden_host.json: writer[*].read_back, generated by make_golden.py)."""
    import io as _io
    from misti_amd import io as mio
    host = json.load(open(os.path.join(GOLDEN, "golden_host.json")))
    assert len(host["writer"]) == 3
    for w in host["writer"]:
        want = w["read_back"]
        f = tmp_path / (w["name"] + ".mi")
        f.write_text(w["text"])
        for src in (str(f), _io.StringIO(w["text"])):
            d = mio.read_migration(src)
            assert d.llh == want["llh"] and d.splitT == want["splitT"] and d.sampleDate == want["sampleDate"] and d.thrh == want["thrh"]
            assert d.jaf == want["jaf"] and d.times == want["times"]
            assert d.lambda1 == want["lambda1"] and d.lambda2 == want["lambda2"] and d.lambdah1 == want["lambdah1"] and d.lambdah2 == want["lambdah2"]
            assert d.migStart is None and d.migEnd is None and d.mi is None
        # round trip: the writer's own text of the same model reads back to the same numbers
        assert len(d.pr) == len(d.times) and any(any(v != 0 for v in row) for row in d.pr)


def test_cli_rejects_gpus_together_with_devices(capsys):
    """`--gpus N` (one rank per GPU) and `--devices` (a device list in one process) exclude each other - N ranks that each
    opened the whole list would run N x D contexts; the command refuses before reading any file."""
    from misti_amd import cli
    rc = cli.main(["a.psmc", "b.psmc", "d.sfs", "20", "--grid-st", "18", "20", "--gpus", "2", "--devices", "0,1"])
    assert rc == 2 and "exclude each other" in capsys.readouterr().err
