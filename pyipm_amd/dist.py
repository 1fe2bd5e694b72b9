"""Multi-GPU host orchestration of the Newton step.

One process per GPU (``torch.distributed``; backend ``nccl`` == RCCL over xGMI).  The KKT
columns are distributed 1-D block-cyclically by panels of ``nb`` columns; a dense
factorisation does not shard into independent units, so there is exactly one real
exchange per panel:

    for p in panels:
        owner(p):   factor panel p (HIP: in-panel updates, 64x64 block pivots, scaling)
                    pack  [-W rows below the panel | the panel's tile inverses]
        everyone:   broadcast(msg, src=owner)                      <- the only data-path collective
        others:     unpack: rebuild the block column L = W * inv(T) locally
        everyone:   rank-nb MFMA update of the columns it owns to the right of p

Sending W (+ 128 KB of tile inverses) instead of W and L halves the bytes on the wire: each
receiver recomputes L with ``nb/64`` small products.

The substitutions move nb-long segments only: the forward sweep sums the segment of panel p over the
ranks (each rank keeps its own share of the running vector), the backward sweep broadcasts each resolved
segment.

Two drivers of the same schedule:

* **native** (the product path for the HIP core): ``pyipm_newton_factor_dist / solve_dist / step_dist``
  run the per-panel loop, the streams and the exchanges inside the library (``csrc/dist_impl.hpp``) -- no
  interpreter between two panels.  This module only binds the exchange: a handle-owned RCCL communicator
  when the process group's backend is ``nccl`` (the 128-byte id travels through ``torch.distributed``), or
  two callbacks that stage through the host when it is ``gloo`` (tests: several ranks sharing one GPU).
* **python** (``native=False``): the loop below over the per-panel C-ABI phases.  ``core`` is then any
  object with the per-panel interface of :class:`pyipm_amd.newton.NewtonCore`; the CPU tests drive it with
  a NumPy model backend over ``gloo`` (world 2/3/4/8), which is how the schedule itself is covered without
  a GPU.
"""
from __future__ import annotations

import numpy as np


class DistNewton(object):
    def __init__(self, core, group=None, stage_through_cpu=None, native=None, p2p=True, serialize=False, selftest=True):
        """p2p / serialize / selftest (callback exchange over gloo only): also install the point-to-point half of the exchange
        (send / recv / all-gather callbacks: the slice messages then travel point to point and the scatter + all-gather panel
        form becomes available), ask the library to issue every operation on its collective stream (what it does for RCCL), and
        run the library's exchange self-test (collective) so that the panel form is decided as comm_init decides it."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.core = core
        self.world, self.rank = int(core.world), int(core.rank)
        self.group = group
        self.native = bool(hasattr(core, "factor_dist")) if native is None else bool(native)
        if self.world > 1:
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed must be initialised for world > 1")
            if dist.get_world_size(group) != self.world or dist.get_rank(group) != self.rank:
                raise RuntimeError("process group does not match the core's (world, rank)")
        backend = dist.get_backend(group) if (self.world > 1) else "none"
        # device tensors cannot ride a gloo group directly: bounce through pinned host memory
        self.stage = (backend == "gloo" and getattr(core, "on_device", True)) if stage_through_cpu is None \
            else bool(stage_through_cpu)
        self.N, self.Npad, self.nb, self.npanels = core.N, core.Npad, core.nb, core.npanels
        self._msg = None
        self.bytes_broadcast = 0
        self.lookahead = True
        self.force_lookahead = False
        # the owner of the next panel factors it on a high-priority side stream while its own share of the
        # bulk update runs on the main stream (HIP cores only; the NumPy model backend has no streams)
        self.overlap_owner = bool(getattr(core, "on_device", True)) and hasattr(core, "sync_stream")
        self._side = None
        self._cb_error = None
        self._p2p, self._serialize, self._selftest = bool(p2p), bool(serialize), bool(selftest)
        if self.native and self.world > 1:
            self._bind_exchange(backend)

    # ------------------------------------------------------------------ native driver: the exchange
    def _bind_exchange(self, backend):
        """nccl: the handle gets its own RCCL communicator (id from rank 0 through the process group);
        gloo: callbacks that stage each buffer through the host (several ranks may share one GPU)."""
        import ctypes
        import os
        from .newton import ALLREDUCE_FN, BCAST_FN, _RawDeviceArray
        torch, dist, core = self.torch, self.dist, self.core
        if backend == "nccl":
            lib = core.lib
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            if os.path.exists(path):
                lib.pyipm_newton_rccl_library(path.encode())            # the RCCL torch itself runs on
            idbuf = (ctypes.c_char * 128)()
            if self.rank == 0:
                rc = lib.pyipm_newton_comm_unique_id(ctypes.cast(idbuf, ctypes.c_void_p))
                if rc:
                    raise RuntimeError("pyipm_newton_comm_unique_id failed (%d)" % rc)
            t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device=core.device)
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            core.comm_init(bytes(t.cpu().tolist()))
            return

        def view(ptr, count):
            return torch.as_tensor(_RawDeviceArray(ptr, count), device=core.device)

        def bcast(user, ptr, nbytes, root, stream):
            try:
                t = view(ptr, nbytes // 8)
                torch.cuda.synchronize(core.device)                     # the library enqueued the producer on `stream`
                h = t.cpu()
                dist.broadcast(h, src=root if self.group is None else dist.get_global_rank(self.group, root), group=self.group)
                if self.rank != root:
                    t.copy_(h)
                    torch.cuda.synchronize(core.device)
                return 0
            except Exception as e:                                      # nothing may propagate through the C frames
                self._cb_error = e
                return 1

        def allreduce(user, ptr, count, op, stream):
            try:
                t = view(ptr, count)
                torch.cuda.synchronize(core.device)
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MAX if op else dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
                torch.cuda.synchronize(core.device)
                return 0
            except Exception as e:
                self._cb_error = e
                return 1

        core.set_exchange(BCAST_FN(bcast), ALLREDUCE_FN(allreduce))
        if not self._p2p:
            return
        from .newton import ALLGATHER_FN, RECV_FN, SEND_FN

        def grank(r):
            return r if self.group is None else dist.get_global_rank(self.group, r)

        def send(user, ptr, nbytes, peer, stream):
            try:
                torch.cuda.synchronize(core.device)
                dist.send(view(ptr, nbytes // 8).cpu(), dst=grank(peer), group=self.group)
                return 0
            except Exception as e:
                self._cb_error = e
                return 1

        def recv(user, ptr, nbytes, peer, stream):
            try:
                h = torch.empty(nbytes // 8, dtype=torch.float64)
                dist.recv(h, src=grank(peer), group=self.group)
                view(ptr, nbytes // 8).copy_(h)
                torch.cuda.synchronize(core.device)
                return 0
            except Exception as e:
                self._cb_error = e
                return 1

        def allgather(user, sptr, rptr, nbytes_per_rank, stream):
            try:
                n = nbytes_per_rank // 8
                torch.cuda.synchronize(core.device)
                h = view(sptr, n).cpu()                                   # (taken before anything is written: send may lie inside recv)
                outs = [torch.empty_like(h) for _ in range(self.world)]
                dist.all_gather(outs, h, group=self.group)
                view(rptr, n * self.world).copy_(torch.cat(outs))
                torch.cuda.synchronize(core.device)
                return 0
            except Exception as e:
                self._cb_error = e
                return 1

        core.set_exchange_p2p(SEND_FN(send), RECV_FN(recv), ALLGATHER_FN(allgather), serialize=self._serialize)
        if self._selftest:
            self._native(core.exchange_selftest)

    # ------------------------------------------------------------------ helpers
    def owner(self, p):
        return p % self.world

    def panel_cols(self, p):
        c0 = p * self.nb
        return c0, min(c0 + self.nb, self.Npad)

    def _bcast(self, t, src):
        if self.world == 1:
            return
        if self.stage:
            h = t.cpu()
            self.dist.broadcast(h, src=src, group=self.group)
            if self.rank != src:
                t.copy_(h)
        else:
            self.dist.broadcast(t, src=src, group=self.group)
        self.bytes_broadcast += t.numel() * t.element_size()

    def _msgbuf(self, numel):
        if self._msg is None or self._msg.numel() < numel:
            self._msg = self.core.new_buffer(max(numel, self.core.panel_msg_numel(0)))
        return self._msg[:numel]

    def _native(self, fn, *args, **kw):
        """Run a native distributed entry point.  An exception raised inside an exchange callback cannot cross the C
        frames: the callback stores it and returns non-zero, the library reports PYIPM_E_COMM.  Here the stored
        exception is re-raised as the cause (and cleared, so it cannot resurface after a later, successful call)."""
        from .newton import NewtonError
        try:
            out = fn(*args, **kw)
        except NewtonError as e:
            cb, self._cb_error = self._cb_error, None
            if cb is not None:
                raise cb from e
            raise
        cb, self._cb_error = self._cb_error, None
        if cb is not None:
            raise cb
        return out

    # ------------------------------------------------------------------ phases
    def factor(self):
        if self.native:
            st = self._native(self.core.factor_dist)
            self.bytes_broadcast = self.core.dist_timings()["bytes"]
            return st
        # world == 1 normally takes the lock-step loop; force_lookahead lets a single rank run the overlapped
        # schedule (side stream + asynchronous broadcasts) so it can be exercised on a one-GPU box
        if self.lookahead and (self.world > 1 or self.force_lookahead):
            return self._factor_lookahead()
        return self._factor_lockstep()

    def _sync_anorm(self):
        """Every rank perturbs alike: the scale of a static pivot is the largest assembled entry over ALL ranks."""
        if self.world > 1 and hasattr(self.core, "anorm"):
            t = self.core.anorm()
            if self.stage:
                h = t.cpu()
                self.dist.all_reduce(h, op=self.dist.ReduceOp.MAX, group=self.group)
                t.copy_(h)
            else:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)

    def _factor_lockstep(self):
        core = self.core
        self._sync_anorm()
        core.factor_begin()
        for p in range(self.npanels):
            own = self.owner(p) == self.rank
            numel = core.panel_msg_numel(p)
            c0, c1 = self.panel_cols(p)
            below = self.Npad - c1
            if own:
                core.factor_panel(p)
            if self.world > 1 and below > 0 and numel > 0:       # numel == 0: nothing to send (slack-block panel)
                buf = self._msgbuf(numel)
                if own:
                    core.panel_pack(p, buf)
                self._bcast(buf, self.owner(p))
                if not own:
                    core.panel_unpack(p, buf)
            if below > 0:
                core.trailing_update(p)
        st = core.factor_end()
        return self._reduce_stats(st)

    def _factor_lookahead(self):
        """One-panel lookahead: as soon as panel p has arrived, the owner of p+1 updates ONLY panel p+1,
        factors it and posts its broadcast asynchronously; every rank then runs the bulk of update p
        while that message is in flight (RCCL runs on its own stream)."""
        core, dist = self.core, self.dist
        self._sync_anorm()
        core.factor_begin()
        np_ = self.npanels
        bufs = [core.new_buffer(core.panel_msg_numel(0)), core.new_buffer(core.panel_msg_numel(0))]

        def post(p):
            """factor (owner) + start the broadcast of panel p; returns (work handle, buffer view)."""
            c0, c1 = self.panel_cols(p)
            own = self.owner(p) == self.rank
            if own:
                core.factor_panel(p)
            if self.Npad - c1 <= 0 or core.panel_msg_numel(p) == 0:   # last panel / slack-block panel: no message
                return None, None
            buf = bufs[p & 1][: core.panel_msg_numel(p)]
            if own:
                core.panel_pack(p, buf)
            if self.stage:                                   # host-staged (test) path: blocking, same ordering
                self._bcast(buf, self.owner(p))
                return None, buf
            if self.world == 1 and not dist.is_initialized():
                return None, buf
            work = dist.broadcast(buf, src=self.owner(p), group=self.group, async_op=True)
            self.bytes_broadcast += buf.numel() * 8
            return work, buf

        torch = self.torch
        side = main = None
        if self.overlap_owner:
            main = torch.cuda.current_stream(core.device)
            if self._side is None:
                self._side = torch.cuda.Stream(device=core.device, priority=-1)
            side = self._side

        def post_owner_overlapped(p):
            """owner of p: factor + pack + start the broadcast on the side stream (ordered after the head
            update just enqueued on the main stream); the caller then enqueues its bulk update on main."""
            side.wait_stream(main)
            with torch.cuda.stream(side):
                core.sync_stream()
                return post(p)

        work, buf = post(0)
        for p in range(np_):
            c0, c1 = self.panel_cols(p)
            if self.Npad - c1 <= 0:
                break
            if work is not None:
                work.wait()                                   # current stream waits for the message
            if side is not None and self.owner(p) == self.rank:
                # the owner factored panel p on the side stream; a panel without a message (slack block) has no
                # work handle to order the main stream behind it
                main.wait_stream(side)
            if self.owner(p) != self.rank and buf is not None:
                core.panel_unpack(p, buf)
            nxt = p + 1
            if nxt < np_:
                if self.owner(nxt) == self.rank:
                    core.trailing_update_range(p, nxt, 1)      # head: bring panel p+1 up to date first
                    if side is not None:
                        work, buf = post_owner_overlapped(nxt)
                        core.sync_stream()                     # back to the main stream
                    else:
                        work, buf = post(nxt)
                else:
                    work, buf = post(nxt)                      # joins the broadcast of p+1 ...
                core.trailing_update_range(p, nxt + 1, np_)    # ... while everyone runs the bulk of update p
            else:
                work, buf = None, None
        if side is not None:
            main.wait_stream(side)                             # the last panel may have been factored there
        st = core.factor_end()
        return self._reduce_stats(st)

    def _reduce_stats(self, st):
        if self.world == 1:
            return st
        torch, dist = self.torch, self.dist
        dev = "cpu" if (self.stage or dist.get_backend(self.group) == "gloo") else self.core.device
        sums = torch.tensor([st["n_neg"], st["n_zero"], st["n_2x2"], st["n_pos"], st["nonfinite"]],
                            dtype=torch.float64, device=dev)
        mx = torch.tensor([st["d_max"], st["growth"], -st["d_min"]], dtype=torch.float64, device=dev)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
        sums, mx = sums.cpu().tolist(), mx.cpu().tolist()
        return {"n_neg": int(sums[0]), "n_zero": int(sums[1]), "n_2x2": int(sums[2]), "n_pos": int(sums[3]),
                "nonfinite": int(sums[4]), "d_max": mx[0], "growth": mx[1], "d_min": -mx[2]}

    def _allreduce_sum(self, t):
        if self.world == 1:
            return
        if self.stage:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def residual(self):
        """g = -grad, complete on every rank."""
        core = self.core
        if self.native:
            return self._native(core.residual_dist)
        g = core.residual()
        if getattr(core, "residual_is_partial", False):     # a HIP rank of several computes the rows it owns
            self._allreduce_sum(g)
        return g

    def solve(self, rhs, flip=True, refine=0):
        """rhs: length-N vector replicated on every rank (device tensor for the HIP core).
        Returns dz replicated on every rank.  Forward: the owner of panel p needs the SUM over the ranks of the
        segment [c0, c1) of their running vectors (each rank pushes the updates of the panels it owns into its own
        vector) -- nb numbers; backward: each resolved segment is broadcast -- nb numbers."""
        core = self.core
        if self.native:
            return self._native(core.solve_dist, rhs, flip=flip, refine=refine)
        if refine:
            raise NotImplementedError("refinement runs in the native driver (pyipm_newton_solve_dist)")
        v = core.new_buffer(self.Npad)
        v.zero_()
        for p in range(self.rank, self.npanels, self.world):          # this rank's share of the right-hand side
            c0, c1 = self.panel_cols(p)
            hi = min(c1, self.N)
            if hi > c0:
                v[c0:hi] = rhs[c0:hi]
        for p in range(self.npanels):                      # forward + block-diagonal
            c0, c1 = self.panel_cols(p)
            own = self.owner(p) == self.rank
            if self.world > 1:
                seg = v[c0:c1].clone()
                self._allreduce_sum(seg)
                self.bytes_broadcast += seg.numel() * 8
                if own:
                    v[c0:c1] = seg
            if own:
                core.fwd_panel(p, v)
                core.diag_panel(p, v)
        for p in range(self.npanels - 1, -1, -1):          # backward
            c0, c1 = self.panel_cols(p)
            if self.owner(p) == self.rank:
                core.bwd_panel(p, v)
            self._bcast(v[c0:c1], self.owner(p))
        dz = v[: self.N].clone()
        if flip and (core.me + core.mi) > 0:
            dz[core.n + core.mi:] *= -1.0                  # pyipm.py:1723-1725
        return dz

    def step(self, delta=0.0, delta_c=0.0, refine=0):
        """residual + assemble + factor + solve + flip (pyipm.py:1717-1725) over all ranks."""
        core = self.core
        if self.native:
            dz, st = self._native(core.step_dist, delta, delta_c, refine=refine)
            self.bytes_broadcast = core.dist_timings()["bytes"]
            return dz, st
        g = self.residual()
        core.assemble(delta, delta_c)
        st = self.factor()
        dz = self.solve(g, flip=True, refine=refine)
        return dz, st
