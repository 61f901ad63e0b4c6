"""misti_set_stream orders a replaced stream before its successor: all batches of a context share its workspaces,
so alternating streams between asynchronous calls must not let batch N+1 overwrite what batch N still reads."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_alternating_streams_on_one_context():
    import torch
    from misti_amd import workloads
    from misti_amd.engine import Engine, truth_spectrum
    w = workloads.config2(lambda *a: truth_spectrum(*a), n_split=16, n_rate=16, first_split=56)
    dev = torch.device("cuda", 0)
    n, P = w.n_cand, w.n_param
    rng = np.random.default_rng(0)
    batches = []
    for k in range(6):                                   # different grids: a batch that reads another's chains gives other numbers
        par = w.params * (1.0 + 0.05 * k)
        batches.append((torch.as_tensor(w.split_time, device=dev), torch.as_tensor(par, device=dev).contiguous()))
    rows = torch.as_tensor(w.jsfs, device=dev).contiguous()
    torch.cuda.synchronize()
    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        want = []
        for s, p in batches:                              # reference results: one batch at a time, synchronised
            out = torch.empty((n, 1), dtype=torch.float64, device=dev)
            e.evaluate_dev(n, s.data_ptr(), p.data_ptr(), 1, rows.data_ptr(), out.data_ptr())
            e.sync()
            want.append(out.cpu().numpy())
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        outs = [torch.empty((n, 1), dtype=torch.float64, device=dev) for _ in batches]
        for rep in range(3):
            for k, (s, p) in enumerate(batches):          # no synchronisation in between, the stream changes every call
                e.use_stream(streams[k % 2].cuda_stream)
                e.evaluate_dev(n, s.data_ptr(), p.data_ptr(), 1, rows.data_ptr(), outs[k].data_ptr())
            torch.cuda.synchronize()
            for k in range(len(batches)):
                assert np.array_equal(outs[k].cpu().numpy(), want[k], equal_nan=True), (rep, k)
        e.use_stream(None)


def test_integer_splits_hint_is_verified_on_the_device():
    """misti_set_hints(MISTI_HINT_INTEGER_SPLITS) on the device-buffer form: the same bits as without it on a batch of whole split times; a
    candidate with a fractional split issued under it is refused (status 4, -inf), its neighbours unchanged; clearing the hint restores it."""
    import torch
    from misti_amd import workloads
    from misti_amd._lib import MistiError
    from misti_amd.engine import Engine, truth_spectrum
    w = workloads.config2(lambda *a: truth_spectrum(*a), n_split=8, n_rate=16, first_split=56)
    dev = torch.device("cuda", 0)
    n = w.n_cand
    split = torch.as_tensor(w.split_time, device=dev)
    frac = split.clone()
    bad = [3, 40, n - 1]
    frac[bad] += 0.25
    par = torch.as_tensor(w.params, device=dev).contiguous()
    rows = torch.as_tensor(w.jsfs, device=dev).contiguous()

    def run(e, s):
        out = torch.empty((n, 1), dtype=torch.float64, device=dev)
        st = torch.empty(n, dtype=torch.int32, device=dev)
        e.evaluate_dev(n, s.data_ptr(), par.data_ptr(), 1, rows.data_ptr(), out.data_ptr(), d_status=st.data_ptr())
        e.sync()
        return out.cpu().numpy()[:, 0], st.cpu().numpy()

    with Engine(w.times, w.lh, **w.engine_kwargs()) as e:
        plain, plain_st = run(e, split)
        plain_frac, plain_frac_st = run(e, frac)
        assert (plain_frac_st[bad] == 0).all() and np.isfinite(plain_frac[bad]).all()
        e.set_hints(integer_splits=True)
        hinted, hinted_st = run(e, split)
        assert np.array_equal(hinted, plain, equal_nan=True) and np.array_equal(hinted_st, plain_st)
        refused, refused_st = run(e, frac)
        assert (refused_st[bad] == 4).all() and np.isneginf(refused[bad]).all()
        keep = np.setdiff1d(np.arange(n), bad)
        assert np.array_equal(refused[keep], plain_frac[keep], equal_nan=True) and np.array_equal(refused_st[keep], plain_frac_st[keep])
        e.set_hints(integer_splits=False)
        again, again_st = run(e, frac)
        assert np.array_equal(again, plain_frac, equal_nan=True) and np.array_equal(again_st, plain_frac_st)
        with pytest.raises(MistiError):
            from misti_amd import _lib
            _lib.check(e._lib.misti_set_hints(e._ctx, 2))
