"""ctypes binding of the C-ABI in ``include/pyipm_newton.h`` — the Newton-step core.

PyTorch-ROCm is plumbing here: it owns the device memory (workspace, staged
blocks, result vectors) and the stream; every kernel is hand-written HIP behind
the C-ABI.  The seam replaced is ``/root/reference/pyipm.py:1717-1725``.

There is NO CPU fallback: if the shared library is missing or no GPU is visible,
constructing ``NewtonCore`` raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_size_t, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PYIPM_NEWTON_LIB") or os.path.join(_HERE, "libpyipm_newton.so")

MEM_DEVICE, MEM_HOST = 0, 1
TILE, PAD = 64, 128

ERRORS = {0: "ok", -1: "bad argument / call order", -2: "HIP runtime error", -3: "workspace too small",
          -4: "NaN/Inf met during factorisation", -5: "no usable HIP device",
          -6: "the exchange failed (RCCL or a caller-supplied callback) or a distributed step timed out"}


class FactorStats(ctypes.Structure):
    """Mirror of ``pyipm_factor_stats``: inertia from block pivots instead of the
    reference's eigen-inertia test (pyipm.py:1378-1381)."""
    _fields_ = [("n_neg", c_int64), ("n_zero", c_int64), ("n_2x2", c_int64), ("n_pos", c_int64),
                ("d_min", c_double), ("d_max", c_double), ("growth", c_double), ("nonfinite", c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class NewtonError(RuntimeError):
    """``code`` is the PYIPM_E_* value; ``stats`` carries the factor statistics when the call produced them
    (PYIPM_E_NONFINITE from factor()/step(): the host may still regularise and retry, as reghess would)."""
    code = None
    stats = None


_lib = None


ABI_VERSION = 6          # PYIPM_NEWTON_ABI_VERSION of include/pyipm_newton.h this file's signatures were written against


def load_library(path: str | None = None):
    """dlopen the HIP core; fails loudly when it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # torch bundles its own libamdhip64; it must be loaded FIRST so this library binds to the same
    # HIP runtime (two runtimes in one process cannot share device pointers or streams).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise NewtonError("HIP library %s not found — run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback." % p)
    lib = ctypes.CDLL(p)
    ctxp = c_void_p
    sig = {
        "pyipm_newton_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int, c_int, c_int]),
        "pyipm_newton_create": (c_int, [POINTER(c_void_p), c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int,
                                        c_void_p, c_size_t, c_void_p]),
        "pyipm_newton_destroy": (c_int, [ctxp]),
        "pyipm_newton_set_stream": (c_int, [ctxp, c_void_p]),
        "pyipm_newton_last_error": (c_char_p, [ctxp]),
        "pyipm_newton_geometry": (c_int, [ctxp, POINTER(c_int64)]),
        "pyipm_newton_stage_blocks": (c_int, [ctxp, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int]),
        "pyipm_newton_stage_vectors": (c_int, [ctxp, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_double, c_double, c_int]),
        "pyipm_newton_residual": (c_int, [ctxp, c_void_p, c_int]),
        "pyipm_newton_assemble": (c_int, [ctxp, c_double, c_double]),
        "pyipm_newton_factor": (c_int, [ctxp, POINTER(FactorStats)]),
        "pyipm_newton_solve": (c_int, [ctxp, c_void_p, c_void_p, c_int, c_int, c_int]),
        "pyipm_newton_solve_info": (c_int, [ctxp, POINTER(c_double)]),
        "pyipm_newton_anorm": (c_int, [ctxp, POINTER(c_void_p)]),
        "pyipm_newton_rcond": (c_int, [ctxp, c_int, c_int, POINTER(c_double)]),
        "pyipm_newton_kkt_matvec": (c_int, [ctxp, c_void_p, c_void_p, c_int]),
        "pyipm_newton_step": (c_int, [ctxp, c_double, c_double, c_int, c_void_p, POINTER(FactorStats), c_int]),
        "pyipm_newton_step_lengths": (c_int, [ctxp, c_double, c_void_p, POINTER(c_double), POINTER(c_double)]),
        "pyipm_newton_merit_info": (c_int, [ctxp, c_void_p, POINTER(c_double)]),
        "pyipm_newton_dots": (c_int, [ctxp, c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), POINTER(c_double)]),
        "pyipm_newton_merit_ray": (c_int, [ctxp, c_void_p, c_double, c_double, POINTER(c_double), POINTER(c_double), c_int,
                                           POINTER(c_double)]),
        "pyipm_newton_factor_panel": (c_int, [ctxp, c_int64]),
        "pyipm_newton_panel_msg_bytes": (c_size_t, [ctxp, c_int64]),
        "pyipm_newton_panel_pack": (c_int, [ctxp, c_int64, c_void_p]),
        "pyipm_newton_panel_unpack": (c_int, [ctxp, c_int64, c_void_p]),
        "pyipm_newton_trailing_update": (c_int, [ctxp, c_int64]),
        "pyipm_newton_trailing_update_range": (c_int, [ctxp, c_int64, c_int64, c_int64]),
        "pyipm_newton_factor_begin": (c_int, [ctxp]),
        "pyipm_newton_factor_end": (c_int, [ctxp, POINTER(FactorStats)]),
        "pyipm_newton_fwd_panel": (c_int, [ctxp, c_int64, c_void_p]),
        "pyipm_newton_diag_panel": (c_int, [ctxp, c_int64, c_void_p]),
        "pyipm_newton_bwd_panel": (c_int, [ctxp, c_int64, c_void_p]),
        "pyipm_newton_kkt_storage": (c_int, [ctxp, POINTER(c_void_p), POINTER(c_int64), POINTER(c_int64)]),
        "pyipm_newton_last_timings": (c_int, [ctxp, POINTER(c_double)]),
        "pyipm_newton_trailing_instances": (c_int, [ctxp, POINTER(c_double)]),
        "pyipm_newton_trailing_bytes": (c_int, [ctxp, POINTER(c_double)]),
        "pyipm_newton_set_option": (c_int, [ctxp, c_char_p, c_double]),
        "pyipm_newton_workspace_bytes_batched": (c_size_t, [c_int64, c_int64, c_int64, c_int]),
        "pyipm_newton_create_batched": (c_int, [POINTER(c_void_p), c_int64, c_int64, c_int64, c_int, c_int,
                                                c_void_p, c_size_t, c_void_p]),
        "pyipm_newton_stage_blocks_batched": (c_int, [ctxp, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                                      c_void_p, c_int64, c_int64]),
        "pyipm_newton_step_batched": (c_int, [ctxp, c_double, c_double, c_void_p, POINTER(FactorStats), c_int]),
        "pyipm_newton_backward_error_batched": (c_int, [ctxp, c_void_p, c_void_p, c_int]),
        "pyipm_newton_stats_batched": (c_int, [ctxp, POINTER(FactorStats)]),
        "pyipm_mfma_f64_peak": (c_int, [c_int, c_int, POINTER(c_double)]),
        "pyipm_newton_block_products": (c_int, [ctxp, c_void_p, c_void_p, c_void_p, c_void_p]),
        "pyipm_newton_block_products_t": (c_int, [ctxp, c_void_p, c_void_p, c_void_p]),
        "pyipm_newton_provider_stats": (c_int, [ctxp, POINTER(c_double)]),
        # distributed driver (dist_impl.hpp)
        "pyipm_newton_set_exchange": (c_int, [ctxp, c_void_p, c_void_p, c_void_p]),
        "pyipm_newton_set_exchange_p2p": (c_int, [ctxp, c_void_p, c_void_p, c_void_p, c_int]),
        "pyipm_newton_exchange_selftest": (c_int, [ctxp]),
        "pyipm_newton_dist_wire": (c_int, [ctxp, POINTER(c_double)]),
        "pyipm_newton_rccl_library": (c_int, [c_char_p]),
        "pyipm_newton_comm_unique_id": (c_int, [c_void_p]),
        "pyipm_newton_workspace_bytes_provider": (c_size_t, [c_int64, c_int64, c_int64]),
        "pyipm_newton_create_provider": (c_int, [POINTER(ctxp), c_int64, c_int64, c_int64, c_int, c_void_p, c_size_t, c_void_p]),
        "pyipm_newton_comm_init": (c_int, [ctxp, c_void_p]),
        "pyipm_newton_comm_ranks": (c_int, [ctxp]),
        "pyipm_newton_comm_bcast_mode": (c_int, [ctxp]),
        "pyipm_newton_owned_rows": (c_int64, [ctxp, POINTER(c_int64)]),
        "pyipm_newton_stage_blocks_owned": (c_int, [ctxp, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int]),
        "pyipm_newton_residual_dist": (c_int, [ctxp, c_void_p, c_int]),
        "pyipm_newton_factor_dist": (c_int, [ctxp, POINTER(FactorStats)]),
        "pyipm_newton_solve_dist": (c_int, [ctxp, c_void_p, c_void_p, c_int, c_int, c_int]),
        "pyipm_newton_kkt_matvec_dist": (c_int, [ctxp, c_void_p, c_void_p, c_int]),
        "pyipm_newton_step_dist": (c_int, [ctxp, c_double, c_double, c_int, c_void_p, POINTER(FactorStats), c_int]),
        "pyipm_newton_dist_timings": (c_int, [ctxp, POINTER(c_double)]),
        "pyipm_newton_abi_version": (c_int, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    lib._pyipm_symbols = tuple(sig)
    if lib.pyipm_newton_abi_version() != ABI_VERSION:      # (an argument list changed under this binding: refuse before any call)
        raise NewtonError("%s speaks interface version %d, this binding was written against %d (include/pyipm_newton.h: "
                          "PYIPM_NEWTON_ABI_VERSION) -- rebuild the library" % (p, lib.pyipm_newton_abi_version(), ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def exported_symbols():
    """Names every entry point ``include/pyipm_newton.h`` declares (used by the CPU tests)."""
    return load_library()._pyipm_symbols


def mfma_f64_peak(device: int = 0, iters: int = 20000) -> float:
    """Measured fp64 MFMA peak (TFLOP/s) from a register-resident MFMA-only loop."""
    out = c_double(0.0)
    rc = load_library().pyipm_mfma_f64_peak(device, iters, ctypes.byref(out))
    if rc:
        raise NewtonError("pyipm_mfma_f64_peak failed: %s" % ERRORS.get(rc, rc))
    return out.value


BCAST_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p)
ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p)
SEND_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p)
RECV_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p)
ALLGATHER_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p)


class _RawDeviceArray(object):
    """fp64 device memory at a raw address, as torch.as_tensor understands it."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": (int(count),), "typestr": "<f8", "version": 2}


class NewtonCore(object):
    """One handle = one KKT system shape (n, me, mi) on one GPU (one rank of ``world``).

    Typical step (the counterpart of pyipm.py:1717-1725)::

        core.stage_blocks(d2L, Je, Ji)             # host-evaluated derivatives, staged once per iteration
        core.stage_vectors(df, ce, ci, s, lda, mu)
        g  = core.residual()                       # -grad                          (:1717)
        core.assemble(delta, delta_c)              # hess + reghess' diagonal shifts (:1718)
        st = core.factor()                         # block LDL' + inertia            (:1378-1381, :1720)
        dz = core.solve(flip=True)                 # substitution + sign flip        (:1720-1725)
    """

    def __init__(self, n, me, mi, device=None, nb=256, world=1, rank=0, provider_only=False):
        """provider_only: block products, residual and kkt_matvec from the staged blocks, no factorisation (O(N) workspace;
        ``pyipm_newton_create_provider``)."""
        import torch
        self.torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise NewtonError("no HIP device visible: the Newton-step core has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.n, self.me, self.mi = int(n), int(me), int(mi)
        self.N = self.n + 2 * self.mi + self.me
        self.nb, self.world, self.rank = int(nb), int(world), int(rank)
        self.provider_only = bool(provider_only)
        if self.provider_only and (self.world != 1 or self.rank != 0):
            raise NewtonError("provider-only handles are single-rank")
        need = self.lib.pyipm_newton_workspace_bytes_provider(self.n, self.me, self.mi) if self.provider_only else \
            self.lib.pyipm_newton_workspace_bytes(self.n, self.me, self.mi, self.nb, self.world, self.rank)
        if need == 0:
            raise NewtonError("invalid geometry n=%d me=%d mi=%d nb=%d world=%d rank=%d" % (n, me, mi, nb, world, rank))
        with torch.cuda.device(self.device):
            self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
            if os.environ.get("PYIPM_POISON_WORKSPACE"):     # test hook (tests/conftest.py): every byte the library does not write
                self.workspace.fill_(255)                    # itself reads back as NaN -- zero pages of a fresh process hide such reads
            h = c_void_p()
            if self.provider_only:
                rc = self.lib.pyipm_newton_create_provider(ctypes.byref(h), self.n, self.me, self.mi, self.device.index,
                                                           c_void_p(self.workspace.data_ptr()), need,
                                                           c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
            else:
                rc = self.lib.pyipm_newton_create(ctypes.byref(h), self.n, self.me, self.mi, self.nb, self.device.index,
                                                  self.world, self.rank, c_void_p(self.workspace.data_ptr()), need,
                                                  c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc:
            raise NewtonError("pyipm_newton_create failed: %s" % ERRORS.get(rc, rc))
        self.h = h
        geo = (c_int64 * 8)()
        self._ck(self.lib.pyipm_newton_geometry(self.h, geo))
        self.Npad, self.npanels, self.ncols_local = int(geo[1]), int(geo[3]), int(geo[4])
        self._keep = {}      # staged device tensors kept alive (the library retains their pointers)

    # -- helpers -----------------------------------------------------------------------------
    def _ck(self, rc, stats=None):
        if rc:
            msg = self.lib.pyipm_newton_last_error(self.h)
            err = NewtonError("%s: %s" % (ERRORS.get(rc, rc), msg.decode() if msg else ""))
            err.code, err.stats = rc, stats
            raise err

    def _dev(self, a, shape=None):
        """Return a contiguous fp64 device tensor for ``a`` (numpy / torch / None)."""
        torch = self.torch
        if a is None:
            return None
        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=torch.float64)
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).to(self.device)
        if shape is not None:
            t = t.reshape(shape)
        return t.contiguous()

    @staticmethod
    def _ptr(t):
        return c_void_p(0) if t is None else c_void_p(t.data_ptr())

    on_device = True

    def new_buffer(self, numel):
        """fp64 device buffer (message / vector container for pyipm_amd.dist)."""
        return self.torch.empty(int(numel), dtype=self.torch.float64, device=self.device)

    def sync_stream(self):
        st = self.torch.cuda.current_stream(self.device).cuda_stream
        self._ck(self.lib.pyipm_newton_set_stream(self.h, c_void_p(st)))
        self._bound_stream = st

    def _use_current_stream(self):
        """The binding allocates outputs / temporaries on torch's CURRENT stream, so the handle must enqueue on that
        same stream: rebind it at the top of every call that touches torch memory (one pointer store)."""
        st = self.torch.cuda.current_stream(self.device).cuda_stream
        if st != getattr(self, "_bound_stream", None):
            self._ck(self.lib.pyipm_newton_set_stream(self.h, c_void_p(st)))
            self._bound_stream = st

    def set_option(self, name, value):
        self._ck(self.lib.pyipm_newton_set_option(self.h, name.encode(), float(value)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.pyipm_newton_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- staging -----------------------------------------------------------------------------
    def stage_blocks(self, d2L, Je=None, Ji=None):
        """d2L (n,n) row-major (upper triangle read), Je (n,me), Ji (n,mi)."""
        self._use_current_stream()
        n, me, mi = self.n, self.me, self.mi
        d2L = self._dev(d2L, (n, n)) if (d2L is not None or not self.provider_only) else None
        Je = self._dev(Je, (n, me)) if me else None
        Ji = self._dev(Ji, (n, mi)) if mi else None
        self._keep.update(d2L=d2L, Je=Je, Ji=Ji)
        self._ck(self.lib.pyipm_newton_stage_blocks(self.h, self._ptr(d2L), n, self._ptr(Je), max(me, 1),
                                                    self._ptr(Ji), max(mi, 1), MEM_DEVICE))

    def stage_vectors(self, df, ce=None, ci=None, s=None, lda=None, mu=0.2, eps=float(np.finfo(np.float64).eps)):
        self._use_current_stream()
        n, me, mi = self.n, self.me, self.mi
        df = self._dev(df, (n,))
        ce = self._dev(ce, (me,)) if me else None
        ci = self._dev(ci, (mi,)) if mi else None
        s = self._dev(s, (mi,)) if mi else None
        lda = self._dev(lda, (me + mi,)) if (me + mi) else None
        self._ck(self.lib.pyipm_newton_stage_vectors(self.h, self._ptr(df), self._ptr(ce), self._ptr(ci),
                                                     self._ptr(s), self._ptr(lda), float(mu), float(eps), MEM_DEVICE))
        self.torch.cuda.current_stream(self.device).synchronize()   # library copied them; temporaries may go

    # -- hot path ------------------------------------------------------------------------------
    def residual(self):
        self._use_current_stream()
        g = self.torch.empty(self.N, dtype=self.torch.float64, device=self.device)
        self._ck(self.lib.pyipm_newton_residual(self.h, self._ptr(g), MEM_DEVICE))
        return g

    def assemble(self, delta=0.0, delta_c=0.0):
        self._use_current_stream()
        self._ck(self.lib.pyipm_newton_assemble(self.h, float(delta), float(delta_c)))

    def factor(self):
        self._use_current_stream()
        st = FactorStats()
        rc = self.lib.pyipm_newton_factor(self.h, ctypes.byref(st))
        self._ck(rc, st.as_dict())
        return st.as_dict()

    def solve(self, rhs=None, flip=True, refine=0):
        """refine > 0: that many refinement steps against the KKT blocks; refine < 0: adaptive (``solve_info``)."""
        self._use_current_stream()
        dz = self.torch.empty(self.N, dtype=self.torch.float64, device=self.device)
        r = None if rhs is None else self._dev(rhs, (self.N,))
        self._ck(self.lib.pyipm_newton_solve(self.h, self._ptr(r), self._ptr(dz), int(bool(flip)), int(refine),
                                             MEM_DEVICE))
        return dz

    def solve_info(self):
        """Outcome of the last solve: refinement steps, backward error before / after, converged flag."""
        o = (c_double * 4)()
        self._ck(self.lib.pyipm_newton_solve_info(self.h, o))
        return {"steps": int(o[0]), "backward_error0": o[1], "backward_error": o[2], "converged": bool(o[3])}

    def rcond(self, it_inv=0, it_pow=0):
        """Estimate of reghess' rcond = min|w| / max|w| (pyipm.py:1379-1381) from the factor and the blocks."""
        self._use_current_stream()
        o = (c_double * 4)()
        self._ck(self.lib.pyipm_newton_rcond(self.h, int(it_inv), int(it_pow), o))
        return {"w_min": o[0], "w_max": o[1], "rcond": o[2], "static_pivot": o[3]}

    def anorm(self):
        """One-element device tensor view of max |assembled entry| (the scale of a static pivot)."""
        ptr = c_void_p()
        self._ck(self.lib.pyipm_newton_anorm(self.h, ctypes.byref(ptr)))
        off = ptr.value - self.workspace.data_ptr()
        return self.workspace[off: off + 8].view(self.torch.float64)

    def matvec(self, v):
        self._use_current_stream()
        v = self._dev(v, (self.N,))
        y = self.torch.empty(self.N, dtype=self.torch.float64, device=self.device)
        self._ck(self.lib.pyipm_newton_kkt_matvec(self.h, self._ptr(v), self._ptr(y), MEM_DEVICE))
        return y

    def step(self, delta=0.0, delta_c=0.0, refine=0):
        """Fused residual + assemble + factor + solve + flip."""
        self._use_current_stream()
        dz = self.torch.empty(self.N, dtype=self.torch.float64, device=self.device)
        st = FactorStats()
        rc = self.lib.pyipm_newton_step(self.h, float(delta), float(delta_c), int(refine), self._ptr(dz),
                                        ctypes.byref(st), MEM_DEVICE)
        self._ck(rc, st.as_dict())
        return dz, st.as_dict()

    def step_lengths(self, tau, dz=None):
        """Fraction-to-the-boundary step lengths (alpha_s, alpha_l) for the direction of the last solve, or for ``dz``
        (device tensor, N doubles, multipliers sign-flipped) when the caller steps along another one."""
        self._use_current_stream()
        a_s, a_l = c_double(1.0), c_double(1.0)
        self._ck(self.lib.pyipm_newton_step_lengths(self.h, float(tau), self._ptr(dz) if dz is not None else None,
                                                    ctypes.byref(a_s), ctypes.byref(a_l)))
        return a_s.value, a_l.value

    MERIT_KEYS = ("ce_l1", "cis_l1", "df_dx", "ds_over_s", "sum_log_s", "kkt_x", "kkt_s", "kkt_ce", "kkt_ci", "comp_sum",
                  "comp_min", "dx_norm", "ds_norm")

    def merit_info(self, dz=None):
        """Merit-function pieces for the staged point and the direction ``dz`` (device tensor; None = the last solve's):
        ``pyipm_newton_merit_info`` -- see include/pyipm_newton.h.  One launch pair, one D2H of 16 doubles."""
        self._use_current_stream()
        out = (c_double * 16)()
        self._ck(self.lib.pyipm_newton_merit_info(self.h, self._ptr(dz) if dz is not None else None, out))
        return dict(zip(self.MERIT_KEYS, out[:13]))

    def dots(self, pairs):
        """[a . b for (a, b) in pairs] for up to 8 pairs of device vectors: one launch, one D2H."""
        self._use_current_stream()
        k = len(pairs)
        A = (c_void_p * k)(*[self._ptr(a) for a, _ in pairs])
        B = (c_void_p * k)(*[self._ptr(b) for _, b in pairs])
        L = (c_int64 * k)(*[int(a.numel()) for a, _ in pairs])
        out = (c_double * k)()
        self._ck(self.lib.pyipm_newton_dots(self.h, k, A, B, L, out))
        return list(out)

    def merit_ray(self, alphas, nu, mu, dz=None, quad=None):
        """phi(x + a dx, s + a ds) - phi(x, s) for every a in ``alphas`` (QP family; ``pyipm_newton_merit_ray``): one launch
        for all candidates, one D2H of len(alphas) doubles."""
        self._use_current_stream()
        k = len(alphas)
        al = (c_double * k)(*[float(a) for a in alphas])
        out = (c_double * k)()
        q = ctypes.byref(c_double(float(quad))) if quad is not None else None
        self._ck(self.lib.pyipm_newton_merit_ray(self.h, self._ptr(dz) if dz is not None else None, float(nu), float(mu), q,
                                                 al, k, out))
        return list(out)

    # -- per-panel phases (used by pyipm_amd.dist) ----------------------------------------------
    def factor_begin(self):
        self._ck(self.lib.pyipm_newton_factor_begin(self.h))

    def factor_end(self):
        st = FactorStats()
        self._ck(self.lib.pyipm_newton_factor_end(self.h, ctypes.byref(st)))
        return st.as_dict()

    def factor_panel(self, p):
        self._ck(self.lib.pyipm_newton_factor_panel(self.h, int(p)))

    def trailing_update(self, p):
        self._ck(self.lib.pyipm_newton_trailing_update(self.h, int(p)))

    def trailing_update_range(self, p, first, count):
        self._ck(self.lib.pyipm_newton_trailing_update_range(self.h, int(p), int(first), int(count)))

    def panel_msg_numel(self, p):
        return int(self.lib.pyipm_newton_panel_msg_bytes(self.h, int(p))) // 8

    def panel_pack(self, p, buf):
        self._ck(self.lib.pyipm_newton_panel_pack(self.h, int(p), self._ptr(buf)))

    def panel_unpack(self, p, buf):
        self._ck(self.lib.pyipm_newton_panel_unpack(self.h, int(p), self._ptr(buf)))

    def fwd_panel(self, p, v):
        self._ck(self.lib.pyipm_newton_fwd_panel(self.h, int(p), self._ptr(v)))

    def diag_panel(self, p, v):
        self._ck(self.lib.pyipm_newton_diag_panel(self.h, int(p), self._ptr(v)))

    def bwd_panel(self, p, v):
        self._ck(self.lib.pyipm_newton_bwd_panel(self.h, int(p), self._ptr(v)))

    # -- QP provider products on the staged blocks ------------------------------
    def block_products(self, v, want=(True, True, True)):
        """(Q v, Je' v, Ji' v) for a device vector v (n); entries not wanted (or of an empty block) are None."""
        self._use_current_stream()
        t = self.torch
        v = self._dev(v, (self.n,))
        mk = lambda k, on: t.empty(k, dtype=t.float64, device=self.device) if (on and k) else None    # noqa: E731
        q, e, i = mk(self.n, want[0]), mk(self.me, want[1]), mk(self.mi, want[2])
        self._ck(self.lib.pyipm_newton_block_products(self.h, self._ptr(v), self._ptr(q), self._ptr(e), self._ptr(i)))
        return q, e, i

    def block_products_t(self, le=None, li=None):
        """Je le + Ji li (n)."""
        self._use_current_stream()
        t = self.torch
        le = self._dev(le, (self.me,)) if (le is not None and self.me) else None
        li = self._dev(li, (self.mi,)) if (li is not None and self.mi) else None
        out = t.empty(self.n, dtype=t.float64, device=self.device)
        self._ck(self.lib.pyipm_newton_block_products_t(self.h, self._ptr(le), self._ptr(li), self._ptr(out)))
        return out

    def provider_stats(self):
        o = (c_double * 4)()
        self._ck(self.lib.pyipm_newton_provider_stats(self.h, o))
        return {"products_ms": o[0], "products_bytes": o[1], "products_t_ms": o[2], "products_t_bytes": o[3]}

    # -- distributed driver (the per-panel schedule runs inside the library; pyipm_amd.dist binds the exchange) -----
    residual_is_partial = property(lambda self: self.world > 1)     # residual() of one rank of several: its share only

    def owned_rows(self):
        """Global indices of the block rows this rank works on (the x-columns it owns), in staging order."""
        n = int(self.lib.pyipm_newton_owned_rows(self.h, None))
        out = (c_int64 * max(n, 1))()
        self.lib.pyipm_newton_owned_rows(self.h, out)
        return np.array(out[:n], dtype=np.int64)

    def stage_blocks_owned(self, d2L_rows, Je_rows=None, Ji_rows=None):
        """Row-sharded staging: only ``owned_rows()`` of d2L (.., n), Je (.., me), Ji (.., mi)."""
        self._use_current_stream()
        n, me, mi = self.n, self.me, self.mi
        r = len(self.owned_rows())
        d2L = self._dev(d2L_rows, (r, n))
        Je = self._dev(Je_rows, (r, me)) if me else None
        Ji = self._dev(Ji_rows, (r, mi)) if mi else None
        self._keep.update(d2L=d2L, Je=Je, Ji=Ji)
        self._ck(self.lib.pyipm_newton_stage_blocks_owned(self.h, self._ptr(d2L), n, self._ptr(Je), max(me, 1),
                                                          self._ptr(Ji), max(mi, 1), MEM_DEVICE))

    def set_exchange(self, bcast_cb, allreduce_cb):
        """ctypes callbacks (BCAST_FN / ALLREDUCE_FN); kept alive with the handle."""
        self._cbs = (bcast_cb, allreduce_cb)
        self._ck(self.lib.pyipm_newton_set_exchange(self.h, ctypes.cast(bcast_cb, c_void_p), ctypes.cast(allreduce_cb, c_void_p), None))

    def set_exchange_p2p(self, send_cb, recv_cb, allgather_cb, serialize=False):
        """ctypes callbacks (SEND_FN / RECV_FN / ALLGATHER_FN) for the point-to-point half of a callback exchange."""
        self._cbs_p2p = (send_cb, recv_cb, allgather_cb)
        self._ck(self.lib.pyipm_newton_set_exchange_p2p(self.h, ctypes.cast(send_cb, c_void_p), ctypes.cast(recv_cb, c_void_p),
                                                        ctypes.cast(allgather_cb, c_void_p), int(bool(serialize))))

    def exchange_selftest(self):
        """COLLECTIVE: agree on and test the scatter + all-gather form of the panel messages on the installed exchange."""
        self._ck(self.lib.pyipm_newton_exchange_selftest(self.h))
        return self.comm_bcast_mode()

    WIRE_KEYS = ("bcast_messages", "bcast_bytes", "sag_messages", "sag_bytes", "p2p_pieces", "allgathers", "stream_hops",
                 "slice_messages", "slice_bytes", "slices_as_broadcast")

    def dist_wire(self):
        t = (c_double * 12)()
        self._ck(self.lib.pyipm_newton_dist_wire(self.h, t))
        d = {k: int(t[i]) for i, k in enumerate(self.WIRE_KEYS)}
        d["chains_with_extra_rows"] = int(t[11])
        return d

    def comm_init(self, id128):
        buf = (ctypes.c_char * 128).from_buffer_copy(bytes(id128))
        self._ck(self.lib.pyipm_newton_comm_init(self.h, ctypes.cast(buf, c_void_p)))

    def comm_ranks(self):
        """Ranks of the handle-owned RCCL communicator as RCCL reports them (0: none)."""
        r = self.lib.pyipm_newton_comm_ranks(self.h)
        if r < 0:
            self._ck(r)
        return int(r)

    def comm_bcast_mode(self):
        """1: the panel messages travel as scatter + all-gather over the handle's communicator, 0: ncclBroadcast."""
        return int(self.lib.pyipm_newton_comm_bcast_mode(self.h))

    def residual_dist(self):
        self._use_current_stream()
        g = self.torch.empty(self.N, dtype=self.torch.float64, device=self.device)
        self._ck(self.lib.pyipm_newton_residual_dist(self.h, self._ptr(g), MEM_DEVICE))
        return g

    def factor_dist(self):
        self._use_current_stream()
        st = FactorStats()
        rc = self.lib.pyipm_newton_factor_dist(self.h, ctypes.byref(st))
        self._ck(rc, st.as_dict())
        return st.as_dict()

    def solve_dist(self, rhs=None, flip=True, refine=0):
        self._use_current_stream()
        dz = self.torch.empty(self.N, dtype=self.torch.float64, device=self.device)
        r = None if rhs is None else self._dev(rhs, (self.N,))
        self._ck(self.lib.pyipm_newton_solve_dist(self.h, self._ptr(r), self._ptr(dz), int(bool(flip)), int(refine), MEM_DEVICE))
        return dz

    def matvec_dist(self, v):
        self._use_current_stream()
        v = self._dev(v, (self.N,))
        y = self.torch.empty(self.N, dtype=self.torch.float64, device=self.device)
        self._ck(self.lib.pyipm_newton_kkt_matvec_dist(self.h, self._ptr(v), self._ptr(y), MEM_DEVICE))
        return y

    def step_dist(self, delta=0.0, delta_c=0.0, refine=0):
        self._use_current_stream()
        dz = self.torch.empty(self.N, dtype=self.torch.float64, device=self.device)
        st = FactorStats()
        rc = self.lib.pyipm_newton_step_dist(self.h, float(delta), float(delta_c), int(refine), self._ptr(dz),
                                             ctypes.byref(st), MEM_DEVICE)
        self._ck(rc, st.as_dict())
        return dz, st.as_dict()

    def trailing_instances(self):
        """Bulk update launches of the last factorisation by kernel instance: {128: {...}, 256: {...}} (profile option on)."""
        t = (c_double * 8)()
        self._ck(self.lib.pyipm_newton_trailing_instances(self.h, t))
        return {bn: {"launches": int(t[4 * k]), "ms": t[4 * k + 1], "flops": t[4 * k + 2], "area": t[4 * k + 3]}   # ("area": algorithmic bytes)
                for k, bn in ((0, 128), (1, 256))}

    def trailing_bytes(self):
        """Algorithmic bytes of the last factorisation's bulk launches in both definitions, by instance:
        {bn: {"c_tiles": .., "c_tiles_and_panels": ..}}."""
        t = (c_double * 4)()
        self._ck(self.lib.pyipm_newton_trailing_bytes(self.h, t))
        return {bn: {"c_tiles": t[k], "c_tiles_and_panels": t[2 + k]} for k, bn in ((0, 128), (1, 256))}

    def dist_timings(self):
        t = (c_double * 8)()
        self._ck(self.lib.pyipm_newton_dist_timings(self.h, t))
        return {"factor_ms": t[0], "chain_ms": t[1], "pack_ms": t[2], "bcast_ms": t[3], "unpack_ms": t[4], "solve_ms": t[5],
                "bytes": int(t[6]), "messages": int(t[7])}

    # -- introspection ---------------------------------------------------------------------------
    def kkt_storage(self):
        """View of the local KKT storage as a (ncols_local, Npad) row-major tensor.  For world=1,
        right after ``assemble`` its upper triangle is the reference's ``triu(H)`` (identity-padded)."""
        ptr, ld, nc = c_void_p(), c_int64(), c_int64()
        self._ck(self.lib.pyipm_newton_kkt_storage(self.h, ctypes.byref(ptr), ctypes.byref(ld), ctypes.byref(nc)))
        ld, nc = int(ld.value), int(nc.value)           # the condensed option stores a smaller matrix
        return self.workspace[: ld * nc * 8].view(self.torch.float64).view(nc, ld)

    def timings(self):
        t = (c_double * 8)()
        self._ck(self.lib.pyipm_newton_last_timings(self.h, t))
        return {"assemble_ms": t[0], "panel_ms": t[1], "trailing_ms": t[2], "solve_ms": t[3],
                "n_trailing": int(t[4]), "trailing_flops": t[5], "factor_ms": t[6], "gram_ms": t[7], "trailing_area": t[7]}
