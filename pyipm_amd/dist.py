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
receiver recomputes L with ``nb/64`` small products.  The substitutions pass the vector
along the owners: after each panel the owner broadcasts the part of the vector it changed.

``core`` is any object with the per-panel interface of :class:`pyipm_amd.newton.NewtonCore`
(the product backend, HIP); the CPU tests drive the same orchestration with a NumPy model
backend over ``gloo``.
"""
from __future__ import annotations

import numpy as np


class DistNewton(object):
    def __init__(self, core, group=None, stage_through_cpu=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.core = core
        self.world, self.rank = int(core.world), int(core.rank)
        self.group = group
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

    # ------------------------------------------------------------------ phases
    def factor(self):
        # world == 1 normally takes the lock-step loop; force_lookahead lets a single rank run the overlapped
        # schedule (side stream + asynchronous broadcasts) so it can be exercised on a one-GPU box
        if self.lookahead and (self.world > 1 or self.force_lookahead):
            return self._factor_lookahead()
        return self._factor_lockstep()

    def _factor_lockstep(self):
        core = self.core
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

    def solve(self, rhs, flip=True):
        """rhs: length-N vector replicated on every rank (device tensor for the HIP core).
        Returns dz replicated on every rank."""
        core = self.core
        v = core.new_buffer(self.Npad)
        v.zero_()
        v[: self.N] = rhs
        for p in range(self.npanels):                      # forward + block-diagonal, owner by owner
            c0, c1 = self.panel_cols(p)
            if self.owner(p) == self.rank:
                core.fwd_panel(p, v)
                core.diag_panel(p, v)
            self._bcast(v[c0:], self.owner(p))
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
        if refine:
            raise NotImplementedError("iterative refinement is single-rank only for now")
        core = self.core
        g = core.residual()
        core.assemble(delta, delta_c)
        st = self.factor()
        dz = self.solve(g, flip=True)
        return dz, st
