"""Overlapped evaluation of independent batches: a pool of *lanes*.

One batch is latency-bound (its longest lambda-correction chain is as sequential as the reference's
solver: ~1.9 ms for the 4 096-point headline grid while 64 of 1 024 SIMDs are busy), so throughput
comes from having many independent batches in flight - bootstrap scans of several data sets, grids
of several models, the vertices of several optimisers.  A lane is one engine context with its own
non-blocking HIP stream (own streams map one-to-one onto hardware queues, 24 per
process; torch's pooled streams share them); batches submitted to different lanes overlap on the GPU,
batches of one lane run in submission order.

    pool = LanePool(times, lh, bands, pulses, n_param=1, cpfit=True, smooth=True, lanes=20)
    tickets = [pool.submit(split_k, params_k, jsfs_k) for k in range(100)]      # device tensors, asynchronous
    results = [t.result() for t in tickets]                                     # (llk, jafs, status) torch tensors

`bench.py` is this loop with timing around it.
"""
from __future__ import annotations

import os

import numpy as np

# the HIP runtime reads this when it initialises: one hardware queue per lane needs more than its default of 4
# (no effect if the runtime is already up - then export it before starting Python)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "22")       # 22: misti_lanes.cpp says why not more

from .engine import Lanes

DEFAULT_LANES = 20          # with 22 hardware queues per process: the null stream and one spare (bench.py)


class Ticket:
    """Result of one submitted batch: device tensors that are valid once `wait()` returns."""

    def __init__(self, lane, llk, jafs, status, event, keep):
        self.lane, self.llk, self.jafs, self.status, self._event, self._keep = lane, llk, jafs, status, event, keep

    def done(self):
        return self._event is None or self._event.query()

    def wait(self):
        if self._event is not None:
            self._event.synchronize()
            self._event = None
        self._keep = None
        return self

    def result(self):
        self.wait()
        return self.llk, self.jafs, self.status


class LanePool:
    """Device tensors in, tickets out, on top of the library's own pool (``misti_create_lanes``: misti_amd.engine.Lanes)."""

    def __init__(self, times, lh, bands=(), pulses=(), n_param=0, lanes=DEFAULT_LANES, device=0, **flags):
        import torch
        self._torch = torch
        self.device = torch.device("cuda", int(device))
        self.n_param = int(n_param)
        self.pool = Lanes(times, lh, bands, pulses, n_param=n_param, device=device, lanes=int(lanes), **flags)
        self.engines = [self.pool.engine(i) for i in range(self.pool.n_lanes)]          # borrowed contexts (per-lane timing, stream handles)
        self.streams = [torch.cuda.ExternalStream(e.stream_handle(), device=self.device) for e in self.engines]
        self._next = 0
        self._outstanding = []

    def close(self):
        if self.pool is not None:
            self.pool.sync()
            self._outstanding = []
            self.streams = []
            self.engines = []
            self.pool.close()
            self.pool = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _dev(self, a, shape):
        t = self._torch
        if a is None:
            return None
        if not isinstance(a, t.Tensor):
            a = t.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape)))
        return a.to(device=self.device, dtype=t.float64).reshape(shape).contiguous()

    def submit(self, split_time, params=None, jsfs=None, lane=None):
        """Queue one batch on the next lane (round-robin) and return at once.  Inputs: NumPy arrays or torch
        tensors (host inputs are copied on the caller's current stream, which the lane then waits for)."""
        t = self._torch
        k = self._next if lane is None else int(lane)
        if lane is None:
            self._next = (self._next + 1) % len(self.engines)
        stream = self.streams[k]
        split = self._dev(split_time, (-1,))
        n = split.numel()
        par = self._dev(params, (n, self.n_param)) if self.n_param else None
        rows = self._dev(jsfs, (-1, 8)) if jsfs is not None else None
        R = 0 if rows is None else rows.shape[0]
        llk = t.empty((n, R), dtype=t.float64, device=self.device)
        jafs = t.empty((n, 7), dtype=t.float64, device=self.device)
        status = t.empty(n, dtype=t.int32, device=self.device)
        ready = t.cuda.Event()
        ready.record(t.cuda.current_stream(self.device))         # inputs (and the allocations above) are ordered on this stream
        stream.wait_event(ready)
        self.pool.evaluate_dev(k, n, split.data_ptr(), par.data_ptr() if par is not None else 0, R, rows.data_ptr() if R else 0,
                               llk.data_ptr() if R else 0, jafs.data_ptr(), 0, 0, status.data_ptr())
        done = t.cuda.Event()
        done.record(stream)
        tk = Ticket(k, llk, jafs, status, done, (split, par, rows))
        # the pool keeps every batch's tensors alive until the batch has run (torch's caching allocator would otherwise
        # hand their memory out again on the caller's stream); record_stream() is not an option - it would make the
        # allocator touch the lane's stream after close() has destroyed it
        self._outstanding = [o for o in self._outstanding if not o.done()]
        self._outstanding.append(tk)
        return tk

    def map(self, batches):
        """Evaluate an iterable of (split_time, params, jsfs) batches, overlapped; results in order as NumPy arrays."""
        tickets = [self.submit(*b) for b in batches]
        out = []
        for tk in tickets:
            llk, jafs, status = tk.result()
            out.append((llk.cpu().numpy(), jafs.cpu().numpy(), status.cpu().numpy()))
        return out
