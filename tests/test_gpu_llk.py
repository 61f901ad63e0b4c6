"""The replicate epilogue as a kernel of its own (misti_llk_dev / more than 8 replicates; MigrationInference.py:600-609, :217-227): every table
shape its launch treats differently - one or two replicate pairs per thread (n_rep <= 512 / above), several replicate tiles (above 1 024), odd
row lengths (scalar stores), fewer candidates than blocks, more candidates than one group per block, candidates without a value - against
the formula in NumPy, and bit for bit against the epilogue fused into the candidate kernel (<= 8 replicates)."""
import io
from math import lgamma

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def grid():
    from misti_amd import synth, io as mio
    return mio.merge_psmc(mio.read_psmc_file(io.StringIO(synth.psmc_text(16, 1, synth.THETA_1))),
                          mio.read_psmc_file(io.StringIO(synth.psmc_text(17, 2, synth.THETA_2))))


def host_llk(jafs, status, jsfs, unfolded):
    d = jsfs[:, 1:]
    if unfolded:
        f, j = d, jafs
    else:
        f = np.stack([d[:, 0] + d[:, 6], d[:, 1] + d[:, 5], d[:, 2] + d[:, 4], d[:, 3]], axis=1)
        j = np.stack([jafs[:, 0] + jafs[:, 6], jafs[:, 1] + jafs[:, 5], jafs[:, 2] + jafs[:, 4], jafs[:, 3]], axis=1)
    const = np.array([lgamma(row.sum() + 1) - sum(lgamma(v + 1) for v in cls) for row, cls in zip(d, f)])
    out = const[None, :] + np.log(j) @ f.T
    out[status != 0] = -np.inf
    return out, np.abs(const)[None, :] + np.abs(np.log(j)) @ f.T


@pytest.mark.parametrize("unfolded", [False, True])
def test_llk_kernel_table_shapes(unfolded):
    import torch
    from misti_amd.engine import Engine
    inp = grid()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11 + unfolded)
    with Engine(inp.times, inp.lambdas, unfolded=unfolded) as e:
        for n_cand, n_rep in ((1, 9), (5, 9), (300, 511), (300, 512), (300, 513), (77, 1000), (2500, 1001), (70, 1024), (33, 1025), (4099, 2049), (70000, 10)):
            jafs = rng.random((n_cand, 7)) + 0.05
            jafs /= jafs.sum(axis=1, keepdims=True)
            status = (rng.random(n_cand) < 0.1).astype(np.int32) * 2
            jsfs = np.zeros((n_rep, 8))
            jsfs[:, 1:] = rng.integers(0, 50000, size=(n_rep, 7))
            jsfs[:, 0] = jsfs[:, 1:].sum(axis=1)
            d_j, d_s, d_r = (torch.as_tensor(a, device=dev) for a in (jafs, status, jsfs))
            out = torch.full((n_cand, n_rep), float("nan"), dtype=torch.float64, device=dev)
            guard = torch.full((64,), 7.0, dtype=torch.float64, device=dev)          # (allocated right behind: an overrun would be a fault or show up in neighbours)
            e.llk_dev(n_cand, d_j.data_ptr(), d_s.data_ptr(), n_rep, d_r.data_ptr(), out.data_ptr())
            e.sync()
            got = out.cpu().numpy()
            want, mag = host_llk(jafs, status, jsfs, unfolded)
            assert not np.isnan(got).any(), (n_cand, n_rep)
            bad = status != 0
            assert np.isneginf(got[bad]).all() and np.isfinite(got[~bad]).all(), (n_cand, n_rep)
            err = np.abs(got[~bad] - want[~bad])
            assert (err <= 1e-13 * mag[~bad] + 1e-9 * 0).all(), (n_cand, n_rep, float((err / mag[~bad]).max()))
            assert float(guard.sum().item()) == 64 * 7.0
            # status NULL: every candidate has a value
            e.llk_dev(n_cand, d_j.data_ptr(), 0, n_rep, d_r.data_ptr(), out.data_ptr())
            e.sync()
            assert np.isfinite(out.cpu().numpy()).all()


def test_separate_kernel_equals_the_fused_epilogue_bit_for_bit():
    """1 000 replicates go through the kernel of their own, their first 8 alone through the epilogue fused into the candidate kernel: same bits."""
    from misti_amd import workloads
    from misti_amd.engine import Engine, truth_spectrum
    w = workloads.config4(lambda *a: truth_spectrum(*a))
    assert w.jsfs.shape[0] >= 1000
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        many = e.evaluate(w.split_time, w.params, w.jsfs)
        few = e.evaluate(w.split_time, w.params, w.jsfs[:8])
    assert np.array_equal(many.llk[:, :8], few.llk, equal_nan=True)
    assert np.array_equal(many.status, few.status)
