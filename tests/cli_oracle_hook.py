"""`python -m cli_oracle_hook <misti_amd.cli arguments>`: the command line of misti_amd.cli with the CPU oracle in place of the HIP engine, for
the gloo runs of tests/test_dist_cpu.py - there is no GPU in the build container, and what those tests rehearse is the rank launch, the
dealing of chains and the gather, not the engine.  The product module has no such switch: this driver replaces `cli._evaluator` from the
outside and names itself as the module `--gpus N` starts.  TEST INFRASTRUCTURE."""
import sys
import warnings
from types import SimpleNamespace

import numpy as np


def oracle_evaluator(a, inp, bands, pulses, k, device):
    from oracle.batch import oracle_eval
    flags = dict(cpfit=a.cpfit, true_eps=a.trueEPS, smooth=not a.nosmooth, unfolded=a.uf)

    def evaluate(split, params, rows):
        rows = np.asarray(rows, dtype=float).reshape(-1, 8)
        llk = np.empty((len(split), rows.shape[0]))
        status = np.zeros(len(split), dtype=np.int32)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for i, s in enumerate(split):
                p = None if params is None else list(params[i])
                v, _, st, _ = oracle_eval(inp.times, inp.lambdas, bands, pulses, flags, inp.sampleDateDiscr, float(s), p, rows, a.mth)
                llk[i], status[i] = v, st
        return SimpleNamespace(llk=llk, status=status, fraction_failed=float((status != 0).mean()))
    return evaluate, (lambda: None)


if __name__ == "__main__":
    from misti_amd import cli
    cli._evaluator = oracle_evaluator
    cli.RANK_MODULE = "cli_oracle_hook"
    sys.exit(cli.main())
