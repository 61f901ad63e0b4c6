"""cpu_baseline leg of bench.py, run as a CHILD PROCESS (TEST INFRASTRUCTURE: the oracle timed as a reported
baseline and as this run's end-to-end checker - never the product path).

    python -m oracle.baseline_child in.npz out.npz

bench.py hands the workload over in a file and starts this module with a fresh interpreter, so the worker pool is
forked by a process that has never initialised HIP, torch or RCCL (forking a process with a live ROCm runtime can
inherit held locks)."""
import json
import sys
from types import SimpleNamespace

import numpy as np

from .batch import oracle_batch


def main():
    src, dst = sys.argv[1], sys.argv[2]
    d = np.load(src)
    meta = json.loads(str(d["meta"]))
    params = d["params"]
    w = SimpleNamespace(times=[float(v) for v in d["times"]], lh=[[float(a), float(b)] for a, b in d["lh"]],
                        bands=[tuple(b) for b in meta["bands"]], pulses=[tuple(p) for p in meta["pulses"]], flags=meta["flags"],
                        sample_date=meta["sample_date"], jsfs=d["jsfs"], split_time=d["split"],
                        params=params if meta["n_param"] else None)
    llk, status, wall = oracle_batch(w, [int(i) for i in d["idx"]], processes=int(meta["cores"]))
    out = dict(llk=llk, status=status, runaway=oracle_batch.last_runaway, wall=wall)
    if "idx_compiled" in d.files:
        # the compiled baseline (C++17 + OpenMP, oracle/cpu/misti_cpu.cpp) on its own (larger) sample of the same workload
        from .cpu_baseline import cpu_eval
        ic = [int(i) for i in d["idx_compiled"]]
        c_llk, _, c_status, c_run, c_wall = cpu_eval(w.times, w.lh, w.bands, w.pulses, w.flags, w.sample_date, w.split_time[ic],
                                                     None if w.params is None else w.params[ic], w.jsfs, int(meta["n_param"]),
                                                     threads=int(meta["cores"]))
        out.update(c_llk=c_llk, c_status=c_status, c_runaway=c_run, c_wall=c_wall)
    np.savez(dst, **out)


if __name__ == "__main__":
    main()
