"""Build libmisti_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmisti_hip.so")
SOURCES = ["misti_kernels.hip", "misti_nm.hip", "misti_api.cpp", "misti_multi.cpp", "misti_lanes.cpp"]
HEADERS = ["misti_device.h", "misti_tables.hpp", "misti_consts.h", os.path.join("..", "..", "include", "misti_hip.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def have_hipcc():
    try:
        hipcc()
        return True
    except RuntimeError:
        return False


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, extra=(), out=None):
    """Compile the kernels and the C-ABI layer into misti_amd/csrc/libmisti_hip.so (or `out`: an experimental
    variant, loaded with MISTI_LIB=<path>; -D switches go in `extra`)."""
    target = out or LIB
    if not force and not out and not stale():
        return LIB
    # Several processes may get here at once (every rank of `bench.py --gpus N` imports the package after a checkout or an edit):
    # one builds, the others wait for the lock and find the library fresh; the compiler writes to a per-process name and the
    # finished file is moved into place, so nobody ever maps a half-written library.
    import fcntl
    lock = open(target + ".lock", "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not out and not stale():
            return LIB
        return _build_locked(target, verbose, extra)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def source_hash(extra=()):
    """sha256 (16 hex digits) over the sources, the headers and the switches a build was made with: the library carries it
    (`misti_build_id()`), stored rocprofv3 counters carry it (profiles/pmc_latest.json), and bench.py refuses to price a run
    with counters of another build."""
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read() + b"\0")
    h.update(" ".join(list(extra) + [os.environ.get("MISTI_FP_CONTRACT", "on"), os.environ.get("MISTI_STAMP", ""), os.environ.get("MISTI_STAMP2", "")]).encode())
    return h.hexdigest()[:16]


def _build_locked(target, verbose, extra):
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
           "-Wall", "-Wno-unused-function", '-DMISTI_BUILD_ID="%s"' % source_hash(extra)]
    # Contraction within a source expression only (clang's `on`), not `fast`: with `fast` the backend fuses a multiply
    # and an add wherever its DAG happens to bring them together, which depends on inlining context - two template
    # instantiations of the same source could then round differently, and a chain's bits would depend on how many
    # chains share its wavefront.  With `on` every expression is fused (or not) as written, in every instantiation.
    cmd += ["-ffp-contract=" + os.environ.get("MISTI_FP_CONTRACT", "on")]
    cmd += list(extra)
    if os.environ.get("MISTI_STAMP2"):         # diagnostic build: stage cycle counts of the spectrum kernel in place of the spectrum
        cmd += ["-DMISTI_STAMP2=1"]
    if os.environ.get("MISTI_STAMP"):          # diagnostic build: per-section cycle stamps in the correction kernel
        cmd += ["-DMISTI_STAMP=1"]
    cmd += ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES]
    tmp = "%s.%d.tmp" % (target, os.getpid())
    cmd += ["-o", tmp, "-ldl", "-lpthread"]            # dlopen: RCCL is bound at first use of the gathered multi-device form
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.run(cmd, check=True, cwd=CSRC)
        os.replace(tmp, target)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return target


if __name__ == "__main__":
    out = None
    if "--out" in sys.argv:
        out = os.path.abspath(sys.argv[sys.argv.index("--out") + 1])
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    print(build(force="--force" in sys.argv, verbose=True, out=out,
                extra=defs + (["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else [])))
