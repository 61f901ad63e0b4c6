"""Measured numbers behind the guards of tests/test_gpu_fullsize.py, test_gpu_configs.py and test_gpu_grid.py (run on the GPU box):

    python tools/measure_guards.py > profiles/rNN_fullsize_contract.txt"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from parity import baseline_contract
from misti_amd import workloads
from misti_amd.engine import Engine, truth_spectrum
import test_gpu_fullsize as tf
spec = lambda *a: truth_spectrum(*a)
print('# checker: the compiled CPU baseline (oracle/cpu/misti_cpu.cpp, the reference\'s algorithm restated, pinned on the reference-generated goldens), values and\n# spreads (eight 2^-48 perturbations per candidate that is not within 1e-9) - not /root/reference itself')
for name, idxf in (("config2", lambda w: np.arange(w.n_cand)), ("config5", lambda w: np.arange(0, w.n_cand, 16)), ("config3", lambda w: np.arange(0, w.n_cand, 4))):
    w = getattr(workloads, name)(spec)
    rep = tf.full_contract(w, idxf(w))
    print(name, "fullsize: both", rep["both"], "tight", rep["tight"], "frac %.4f" % (rep["tight"] / rep["both"]), "self", rep["self_bound"], "outside", len(rep["outside"]),
          [(int(k), float(rep["rel"][k]), float(rep["run"][k])) for k in rep["outside"]], "mismatch", len(rep["mismatch"]), "worst_tight %.3g" % rep["worst_tight"])
# sample tests (oracle-sized samples, all replicates of interest)
for name, n_sample in (("config2", 96), ("config3", 64), ("config4", 24), ("config5", 64)):
    w = getattr(workloads, name)(spec)
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        res = e.evaluate(w.split_time, w.params, w.jsfs)
    idx = np.linspace(0, w.n_cand - 1, n_sample).astype(int)
    rep = baseline_contract(w, idx, res.llk, res.status)
    print(name, "sample: both", rep["both"], "tight", rep["tight"], "self", rep["self_bound"], "outside", len(rep["outside"]),
          [(int(k), float(rep["rel"][k]), float(rep["run"][k])) for k in rep["outside"]], "mismatch", len(rep["mismatch"]), "runaway", int((rep["run"] >= 5).sum()))
