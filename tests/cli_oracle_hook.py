"""Test hook of misti_amd.cli (MISTI_TEST_EVALUATOR=cli_oracle_hook:make): the batch evaluator of grid mode with the CPU oracle
behind it, for the gloo runs of tests/test_dist_cpu.py - there is no GPU in the build container.  TEST INFRASTRUCTURE."""
import warnings
from types import SimpleNamespace

import numpy as np


def make(times, lambdas, bands, pulses, n_param, flags, sample_date, mixture_th):
    from oracle.batch import oracle_eval

    def evaluate(split, params, rows):
        rows = np.asarray(rows, dtype=float).reshape(-1, 8)
        llk = np.empty((len(split), rows.shape[0]))
        status = np.zeros(len(split), dtype=np.int32)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for i, s in enumerate(split):
                p = None if params is None else list(params[i])
                v, _, st, _ = oracle_eval(times, lambdas, bands, pulses, flags, sample_date, float(s), p, rows, mixture_th)
                llk[i], status[i] = v, st
        return SimpleNamespace(llk=llk, status=status, fraction_failed=float((status != 0).mean()))
    return evaluate
