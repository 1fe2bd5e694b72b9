"""Batched independent Newton steps (BASELINE.json configs[4]: 512 independent n=256 QPs, multi-start).

Independent problems shard with no exchange at all ("replicas only"), and a system
this small (Npad <= 1024) fits one workgroup: the library's batched handle
(``pyipm_newton_create_batched`` / ``stage_blocks_batched`` / ``step_batched``, ``csrc/kernels_batched.hpp``)
runs the whole batch with four launches -- residuals, assembly, ONE factorisation launch with a workgroup per
problem (2 resident per CU: 512 problems in flight on an MI355X), substitutions.  Across GPUs the batch is
simply split by rank.  Same algorithm and tile-inversion code as the single-system path; no CPU fallback."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_void_p

import numpy as np

from .newton import ERRORS, MEM_DEVICE, FactorStats, NewtonError, load_library


class LazyStats(object):
    """The per-problem factor statistics of one ``step_all`` -- a list of dicts, read from the device at first use (``len``,
    indexing, iteration).  Valid until the handle's next step."""

    def __init__(self, owner, step_id):
        self._owner, self._step_id, self._data = owner, step_id, None

    def _fetch(self):
        if self._data is None:
            self._data = self._owner._fetch_stats(self._step_id)
        return self._data

    def __len__(self):
        return len(self._fetch())

    def __getitem__(self, i):
        return self._fetch()[i]

    def __iter__(self):
        return iter(self._fetch())

    def __eq__(self, other):
        return list(self._fetch()) == list(other)

    def __repr__(self):
        return repr(self._fetch())


class BatchedNewton(object):
    condensed_tol = 1e-9                   # backward-error bar (against the FULL blocks) a condensed direction must meet

    def __init__(self, n, me, mi, batch=None, device=None, workers=None, nb=None, refine=0, condensed=False, guard=True):
        """``workers`` / ``nb`` are accepted and ignored (an earlier version drove one handle per host thread).
        ``condensed``: factor the condensed system of every problem (n + me + |active rows| columns instead of n + 2 mi + me:
        the (s, lambda_i) pairs eliminated analytically, SURVEY 8f rank 2 for the batched handle) -- same directions, inertia
        of the full matrix.  With ``guard`` every condensed step is checked on the device (inertia, static pivots, backward
        error against the full blocks: ``pyipm_newton_backward_error_batched``) and the batch is redone with the full form
        when a problem misses ``condensed_tol`` (counted in ``n_condensed_fallback``)."""
        import torch
        if refine:
            raise NotImplementedError("iterative refinement is not part of the batched path")
        self.condensed, self.guard = bool(condensed and mi), bool(guard)
        self.n_condensed_fallback, self.last_backward_errors = 0, None
        if self.condensed:
            self._opts = {"condensed": 1.0}
        self.torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise NewtonError("no HIP device visible: the Newton-step core has no CPU fallback")
        self.n, self.me, self.mi = int(n), int(me), int(mi)
        self.N = self.n + 2 * self.mi + self.me
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.h, self.batch, self.workspace = None, 0, None
        self._keep = None
        if batch:
            self._create(int(batch))

    def _create(self, batch):
        torch = self.torch
        self.close()
        need = self.lib.pyipm_newton_workspace_bytes_batched(self.n, self.me, self.mi, batch)
        if need == 0:
            raise NewtonError("batched handles need n + 2 mi + me <= 1024 (got %d)" % self.N)
        with torch.cuda.device(self.device):
            self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
            if os.environ.get("PYIPM_POISON_WORKSPACE"):     # test hook (tests/conftest.py): every byte the library does not write
                self.workspace.fill_(255)                    # itself reads back as NaN -- zero pages of a fresh process hide such reads
            h = c_void_p()
            rc = self.lib.pyipm_newton_create_batched(ctypes.byref(h), self.n, self.me, self.mi, batch, self.device.index,
                                                      c_void_p(self.workspace.data_ptr()), need,
                                                      c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc:
            raise NewtonError("pyipm_newton_create_batched failed: %s" % ERRORS.get(rc, rc))
        self.h, self.batch = h, batch
        self._step_id = getattr(self, "_step_id", 0) + 1    # (statistics of a step on the old handle are gone)
        for k, v in getattr(self, "_opts", {}).items():
            self._ck(self.lib.pyipm_newton_set_option(self.h, k.encode(), v))

    def set_option(self, name, value):
        """Same options as ``NewtonCore.set_option`` where they apply (e.g. ``tile_blocked``); kept for the handle that the
        first ``step_all`` creates (its size is the batch's)."""
        if not hasattr(self, "_opts"):
            self._opts = {}
        self._opts[name] = float(value)
        if getattr(self, "h", None):
            self._ck(self.lib.pyipm_newton_set_option(self.h, name.encode(), float(value)))

    def _ck(self, rc):
        if rc:
            msg = self.lib.pyipm_newton_last_error(self.h)
            raise NewtonError("%s: %s" % (ERRORS.get(rc, rc), msg.decode() if msg else ""))

    def step_all(self, d2L, Je=None, Ji=None, df=None, ce=None, ci=None, s=None, lda=None, mu=0.2,
                 delta=0.0, delta_c=0.0, eps=float(np.finfo(np.float64).eps)):
        """All arguments carry a leading batch dimension (torch device tensors or NumPy arrays):
        d2L (B,n,n), Je (B,n,me), Ji (B,n,mi), df (B,n), ce (B,me), ci (B,mi), s (B,mi), lda (B,me+mi).
        Returns (dz (B,N) device tensor, list of per-problem factor statistics)."""
        torch = self.torch
        n, me, mi = self.n, self.me, self.mi

        def dev(a, shape):
            if a is None:
                return None
            if isinstance(a, torch.Tensor):
                t = a.to(device=self.device, dtype=torch.float64)
            else:
                t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).to(self.device)
            return t.reshape(shape).contiguous()

        B = int(d2L.shape[0])
        if self.h is None or B != self.batch:
            self._create(B)
        blocks = (dev(d2L, (B, n, n)), dev(Je, (B, n, me)) if me else None, dev(Ji, (B, n, mi)) if mi else None)
        vecs = (dev(df, (B, n)), dev(ce, (B, me)) if me else None, dev(ci, (B, mi)) if mi else None,
                dev(s, (B, mi)) if mi else None, dev(lda, (B, me + mi)) if (me + mi) else None)
        self._keep = (blocks, vecs)            # the library retains the block pointers

        def ptr(t):
            return c_void_p(0) if t is None else c_void_p(t.data_ptr())

        self._ck(self.lib.pyipm_newton_set_stream(self.h, c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        self._ck(self.lib.pyipm_newton_stage_blocks_batched(self.h, ptr(blocks[0]), n, n * n, ptr(blocks[1]), me, n * me,
                                                            ptr(blocks[2]), mi, n * mi))
        self._ck(self.lib.pyipm_newton_stage_vectors(self.h, ptr(vecs[0]), ptr(vecs[1]), ptr(vecs[2]), ptr(vecs[3]),
                                                     ptr(vecs[4]), float(mu), float(eps), MEM_DEVICE))
        out = torch.empty((B, self.N), dtype=torch.float64, device=self.device)
        # the step is five launches and returns once they are enqueued; the B statistics records stay on the device until somebody
        # looks at them (LazyStats: round 6 -- copying and unpacking 512 records per step was a third of the step's wall time)
        self._ck(self.lib.pyipm_newton_step_batched(self.h, float(delta), float(delta_c), ptr(out), None, MEM_DEVICE))
        self._step_id += 1
        stats = LazyStats(self, self._step_id)
        if self._opts_get("condensed") and self.guard and mi:
            # the condensed form's guard (as HipNewtonBackend's for a single system): right inertia, no static pivot, and a
            # direction that satisfies the FULL blocks; otherwise the batch is redone with the full 4-block system
            be = self.backward_errors(out)
            self.last_backward_errors = be
            bad = (be > self.condensed_tol) | ~torch.isfinite(be)
            ok = not bool(bad.any()) and all(x["n_neg"] == me + mi and x["n_zero"] == 0 and not x["nonfinite"] for x in stats)
            if not ok:
                self.n_condensed_fallback += 1
                self._ck(self.lib.pyipm_newton_set_option(self.h, b"condensed", 0.0))
                try:
                    self._ck(self.lib.pyipm_newton_step_batched(self.h, float(delta), float(delta_c), ptr(out), None, MEM_DEVICE))
                    self._step_id += 1
                    stats = LazyStats(self, self._step_id)
                    stats._fetch()                      # (read while the handle is in the full form)
                finally:
                    self._ck(self.lib.pyipm_newton_set_option(self.h, b"condensed", 1.0))
        return out, stats

    def _fetch_stats(self, step_id):
        if step_id != self._step_id:
            raise NewtonError("the statistics of an earlier step_all are gone: the handle keeps the last step's only -- read them "
                              "(len / index / iterate) before the next step")
        st = (FactorStats * self.batch)()
        rc = self.lib.pyipm_newton_stats_batched(self.h, st)
        if rc and rc != -4:                                   # (-4 = PYIPM_E_NONFINITE: the records say which problems)
            self._ck(rc)
        return [x.as_dict() for x in st]

    def _opts_get(self, name):
        return getattr(self, "_opts", {}).get(name, 0.0)

    def backward_errors(self, dz):
        """|g - Hc raw| / |g| per problem for the directions ``dz`` (B, N) of the last step, Hc from the staged blocks (device
        tensor of B doubles)."""
        torch = self.torch
        be = torch.empty(self.batch, dtype=torch.float64, device=self.device)
        self._ck(self.lib.pyipm_newton_backward_error_batched(self.h, c_void_p(dz.data_ptr()), c_void_p(be.data_ptr()), MEM_DEVICE))
        return be

    def last_ms(self):
        """HIP-event times of the last step (ms): residual + assembly, factorisation, substitutions, the whole step."""
        from ctypes import c_double
        t = (c_double * 8)()
        self._ck(self.lib.pyipm_newton_last_timings(self.h, t))
        return {"assemble_ms": t[0], "factor_ms": t[6], "solve_ms": t[3], "step_ms": t[1]}

    def close(self):
        if getattr(self, "h", None):
            self.lib.pyipm_newton_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
