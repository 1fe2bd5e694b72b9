"""Batched independent Newton steps (BASELINE.json configs[4]: 512 independent n=256 QPs, multi-start).

Independent problems shard with no exchange at all ("replicas only"): a pool of
``workers`` handles, each on its own HIP stream and driven by its own host thread (ctypes releases the
GIL inside the C-ABI calls), works through the batch.  Across GPUs the batch is simply split by rank.
Every problem goes through the same HIP path as the single-system case (no CPU fallback)."""
from __future__ import annotations

import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np


class BatchedNewton(object):
    def __init__(self, n, me, mi, device=None, workers=8, nb=128, refine=0):
        import torch
        self.torch = torch
        self.n, self.me, self.mi = int(n), int(me), int(mi)
        self.N = self.n + 2 * self.mi + self.me
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.workers, self.nb, self.refine = int(workers), int(nb), int(refine)
        self._local = threading.local()
        self._pool = ThreadPoolExecutor(max_workers=self.workers)
        self._cores = []

    def _core(self):
        loc = self._local
        if not hasattr(loc, "core"):
            from .newton import NewtonCore
            torch = self.torch
            torch.cuda.set_device(self.device)
            loc.stream = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(loc.stream):
                loc.core = NewtonCore(self.n, self.me, self.mi, device=self.device.index, nb=self.nb)
            self._cores.append(loc.core)
        return loc.core, loc.stream

    def _one(self, b, blocks, vecs, mu, delta, delta_c, out):
        torch = self.torch
        core, stream = self._core()
        d2L, Je, Ji = blocks
        df, ce, ci, s, lda = vecs
        with torch.cuda.stream(stream):
            core.stage_blocks(d2L[b], None if Je is None else Je[b], None if Ji is None else Ji[b])
            core.stage_vectors(df[b], None if ce is None else ce[b], None if ci is None else ci[b],
                               None if s is None else s[b], None if lda is None else lda[b], mu=mu)
            dz, st = core.step(delta, delta_c, refine=self.refine)
            out[b].copy_(dz)
            stream.synchronize()
        return st

    def step_all(self, d2L, Je=None, Ji=None, df=None, ce=None, ci=None, s=None, lda=None, mu=0.2,
                 delta=0.0, delta_c=0.0):
        """All arguments carry a leading batch dimension (torch device tensors or NumPy arrays):
        d2L (B,n,n), Je (B,n,me), Ji (B,n,mi), df (B,n), ce (B,me), ci (B,mi), s (B,mi), lda (B,me+mi).
        Returns (dz (B,N) device tensor, list of per-problem factor statistics)."""
        torch = self.torch

        def dev(a):
            if a is None:
                return None
            if isinstance(a, torch.Tensor):
                return a.to(device=self.device, dtype=torch.float64)
            return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).to(self.device)

        blocks = (dev(d2L), dev(Je), dev(Ji))
        vecs = (dev(df), dev(ce), dev(ci), dev(s), dev(lda))
        B = blocks[0].shape[0]
        out = torch.empty((B, self.N), dtype=torch.float64, device=self.device)
        torch.cuda.synchronize(self.device)
        futs = [self._pool.submit(self._one, b, blocks, vecs, mu, delta, delta_c, out) for b in range(B)]
        stats = [f.result() for f in futs]
        return out, stats

    def close(self):
        self._pool.shutdown(wait=True)
        for c in self._cores:
            c.close()
