"""Child process of tests/test_gpu_multi.py::test_gathered_form_on_three_contexts_through_the_rccl_double (run with MISTI_RCCL_LIB naming
the double built from tests/multi_host/fake_rccl.cpp with hipcc): misti_multi_eval_batch_dev on the device list {0, 0, 0} - three
contexts, three persistent workers, one grouped ncclAllGather over three communicators - with ragged and empty shards, against one
context's misti_eval_batch, bit for bit; the status table; a worker that throws.  Prints `key = value` lines."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from misti_amd import workloads
    from misti_amd._lib import MistiError
    from misti_amd.engine import Engine, MultiEngine, truth_spectrum
    w = workloads.config2(lambda *a: truth_spectrum(*a), n_split=8, n_rate=6, first_split=58, max_rate=0.5)
    w.jsfs = np.vstack([w.jsfs, w.jsfs * 0.5, w.jsfs * 0.25])
    n, R, D = w.n_cand, 3, 3
    dev = torch.device("cuda", 0)
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        one = e.evaluate(w.split_time, w.params, w.jsfs)
    d_jsfs = torch.as_tensor(w.jsfs, device=dev)
    identical, rounds = 1, 0
    with MultiEngine(w.times, w.lh, devices=(0,) * D, **w.engine_kwargs()) as m:
        # shard layouts: ragged, one empty, all in one, one candidate each
        for cuts in ([0, 20, 33, n], [0, 0, 30, n], [0, n, n, n], [0, 1, 2, 3], [0, 17, 17, n]):
            idx = [np.arange(cuts[d], cuts[d + 1]) for d in range(D)]
            per = max(1, max(len(i) for i in idx)) + 2
            d_split = [torch.as_tensor(w.split_time[i] if len(i) else np.zeros(1), device=dev) for i in idx]
            d_par = [torch.as_tensor(w.params[i] if len(i) else np.zeros((1, w.n_param)), device=dev).contiguous() for i in idx]
            tables = [torch.full((D, per, R), 7.0, dtype=torch.float64, device=dev) for _ in range(D)]
            stats = [torch.full((D, per), 55, dtype=torch.int32, device=dev) for _ in range(D)]
            torch.cuda.synchronize()
            for _ in range(2):                       # the communicators are made once and reused; a second batch overwrites the first in place
                m.evaluate_dev_gathered([len(i) for i in idx], per, [t.data_ptr() for t in d_split], [t.data_ptr() for t in d_par], R,
                                        [d_jsfs.data_ptr()] * D, [t.data_ptr() for t in tables], [t.data_ptr() for t in stats])
                m.sync()
            rounds += 1
            for k in range(D):                       # EVERY context's table is the whole table
                got, st = tables[k].cpu().numpy(), stats[k].cpu().numpy()
                for d in range(D):
                    nd = len(idx[d])
                    ok = (np.array_equal(got[d, :nd], one.llk[idx[d]], equal_nan=True) and np.isnan(got[d, nd:]).all()
                          and np.array_equal(st[d, :nd], one.status[idx[d]]) and (st[d, nd:] == -1).all())
                    if not ok:
                        identical = 0
                        print("mismatch: cuts %s table %d block %d" % (cuts, k, d))
        hook = m._lib.misti_multi_test_throw_in_worker_
        hook.restype, hook.argtypes = C.c_int, [C.c_void_p, C.c_int]
        hook(m._m, 1)
        try:
            m.evaluate_dev_gathered([len(i) for i in idx], per, [t.data_ptr() for t in d_split], [t.data_ptr() for t in d_par], R,
                                    [d_jsfs.data_ptr()] * D, [t.data_ptr() for t in tables], [t.data_ptr() for t in stats])
            print("throw = not reported")
        except MistiError as ex:
            print("throw = %s" % ("reported" if "context 1 of 3" in str(ex) else str(ex)))
        hook(m._m, -1)
        m.sync()
    double = C.CDLL(os.environ["MISTI_RCCL_LIB"])
    print("collectives = %d" % double.misti_test_rccl_double_collectives())
    print("rounds = %d" % rounds)
    print("identical = %d" % identical)
    print("finite = %d" % int(np.isfinite(one.llk).sum()))


if __name__ == "__main__":
    main()
