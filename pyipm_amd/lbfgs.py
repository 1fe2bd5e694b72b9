"""ctypes binding of ``include/pyipm_lbfgs.h`` — the L-BFGS search direction on the device.

Counterpart of the reference's compiled ``lbfgs_dir_func`` (``/root/reference/pyipm.py:872-875``, built by
``lbfgs_builder`` ``:1007-1182``, called from ``lbfgs_dir`` ``:1184-1246``).  Same shared object and the
same conventions as ``pyipm_amd.newton``; PyTorch owns tensors and the stream, every kernel is HIP.
No CPU fallback: without the library or a GPU the constructor raises.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_size_t, c_void_p

import numpy as np

from .newton import ERRORS, MEM_DEVICE, MEM_HOST, NewtonError, load_library


class LbfgsStats(ctypes.Structure):
    """Mirror of ``pyipm_lbfgs_stats``."""
    _fields_ = [("m", c_int64), ("n_neg", c_int64), ("n_zero", c_int64), ("regularised", c_int64),
                ("n_factor", c_int64), ("d_min", c_double), ("d_max", c_double), ("small_pivot_min", c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_SIG = {
    "pyipm_lbfgs_create": (c_int, [POINTER(c_void_p), c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "pyipm_lbfgs_destroy": (c_int, [c_void_p]),
    "pyipm_lbfgs_last_error": (c_char_p, [c_void_p]),
    "pyipm_lbfgs_set_stream": (c_int, [c_void_p, c_void_p]),
    "pyipm_lbfgs_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int, c_int]),
    "pyipm_lbfgs_stage_jacobian": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int]),
    "pyipm_lbfgs_direction": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int, c_void_p, c_int64,
                                      c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_double, c_double,
                                      c_void_p, c_int, c_int, POINTER(LbfgsStats)]),
    "pyipm_lbfgs_last_timings": (c_int, [c_void_p, POINTER(c_double)]),
    "pyipm_lbfgs_set_option": (c_int, [c_void_p, c_char_p, c_double]),
    "pyipm_lbfgs_set_allreduce": (c_int, [c_void_p, c_void_p, c_void_p]),
}
ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_int64, c_void_p)


class _RawDeviceArray(object):
    """fp64 device memory owned by the library, exposed to torch without a copy."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": (int(count),), "typestr": "<f8",
                                         "version": 2, "strides": None}
_bound = None


def load():
    global _bound
    if _bound is None:
        lib = load_library()
        for name, (res, args) in _SIG.items():
            fn = getattr(lib, name)              # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        _bound = lib
    return _bound


def exported_symbols():
    """Names every entry point ``include/pyipm_lbfgs.h`` declares (used by the CPU tests)."""
    load()
    return tuple(_SIG)


class LbfgsCore(object):
    """One handle = one problem shape (n, me, mi) and a bound on the stored pairs.

    ::

        core.stage_jacobian(Je, Ji)                               # dce(x), dci(x); once if the constraints are linear
        dz, st = core.direction(g, s, lda, zeta, S, Y, SS, L, D, reg=reg)     # RAW direction (flip=False), :1713
    """

    def __init__(self, n, me, mi, max_pairs, device=None, nb=256, group=None, shard=False):
        """shard=True (or a process group): this handle holds ``n`` ROWS of a row-sharded problem; the three sums over
        the ranks (J'J, J'[g_x|W], W'[g_x|W]) go through ``torch.distributed.all_reduce`` on ``group`` — backend "nccl"
        (= RCCL) in place on the device buffers, "gloo" staged through the host (tests)."""
        import torch
        self.torch = torch
        self.lib = load()
        if not torch.cuda.is_available():
            raise NewtonError("no HIP device visible: the L-BFGS direction has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.n, self.me, self.mi, self.cap = int(n), int(me), int(mi), int(max_pairs)
        self.N = self.n + 2 * self.mi + self.me
        if self.lib.pyipm_lbfgs_workspace_bytes(self.n, self.me, self.mi, self.cap, int(nb)) == 0:
            raise NewtonError("invalid L-BFGS geometry n=%d me=%d mi=%d max_pairs=%d nb=%d" % (n, me, mi, max_pairs, nb))
        h = c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.pyipm_lbfgs_create(ctypes.byref(h), self.n, self.me, self.mi, self.cap, int(nb),
                                             self.device.index,
                                             c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc:
            raise NewtonError("pyipm_lbfgs_create failed: %s" % ERRORS.get(rc, rc))
        self.h = h
        self.group, self.bytes_reduced, self._cb = group, 0, None
        if shard or group is not None:
            self._install_allreduce()

    def _install_allreduce(self):
        import torch.distributed as dist
        torch = self.torch
        if not dist.is_initialized():
            raise NewtonError("row-sharded L-BFGS needs an initialised torch.distributed process group")
        staged = dist.get_backend(self.group) == "gloo"

        def cb(user, ptr, count, stream):
            try:
                t = torch.as_tensor(_RawDeviceArray(ptr, count), device=self.device)
                if staged:
                    hbuf = t.cpu()                                   # synchronises the stream
                    dist.all_reduce(hbuf, op=dist.ReduceOp.SUM, group=self.group)
                    t.copy_(hbuf)
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)     # ordered on the current stream
                self.bytes_reduced += 8 * int(count)
                return 0
            except Exception as e:                                   # nothing may propagate through the C frames
                self._cb_error = e
                return 1

        self._cb = ALLREDUCE_FN(cb)                                  # keep alive as long as the handle
        self._ck(self.lib.pyipm_lbfgs_set_allreduce(self.h, ctypes.cast(self._cb, c_void_p), None))

    def _ck(self, rc):
        if rc:
            msg = self.lib.pyipm_lbfgs_last_error(self.h)
            err = NewtonError("%s: %s" % (ERRORS.get(rc, rc), msg.decode() if msg else ""))
            cause = getattr(self, "_cb_error", None)          # what the all-reduce callback swallowed, if that was it
            self._cb_error = None
            if cause is not None:
                raise err from cause
            raise err

    def _dev(self, a, shape):
        torch = self.torch
        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=torch.float64)
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).to(self.device)
        return t.reshape(shape).contiguous()

    @staticmethod
    def _ptr(t):
        return c_void_p(0) if t is None else c_void_p(t.data_ptr())

    def set_option(self, name, value):
        self._ck(self.lib.pyipm_lbfgs_set_option(self.h, name.encode(), float(value)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.pyipm_lbfgs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stage_jacobian(self, Je=None, Ji=None):
        """Je (n, me), Ji (n, mi): NumPy arrays or device tensors (copied into the padded operand buffer)."""
        n, me, mi = self.n, self.me, self.mi
        if me + mi == 0:
            return
        Je = self._dev(Je, (n, me)) if me else None
        Ji = self._dev(Ji, (n, mi)) if mi else None
        self._ck(self.lib.pyipm_lbfgs_stage_jacobian(self.h, self._ptr(Je), max(me, 1), self._ptr(Ji), max(mi, 1),
                                                     MEM_DEVICE))
        self.torch.cuda.current_stream(self.device).synchronize()      # Je / Ji temporaries may die now

    def direction(self, g, s, lda, zeta, S, Y, SS, L, D, reg=0.0, eps=float(np.finfo(np.float64).eps), flip=False):
        """Returns (dz device tensor of length n + 2 mi + me, stats dict)."""
        torch = self.torch
        n, me, mi, N = self.n, self.me, self.mi, self.N
        m = 0 if S is None else int(S.shape[1])
        g = self._dev(g, (N,))
        s = self._dev(s, (mi,)) if mi else None
        lda = self._dev(lda, (me + mi,)) if (me + mi) else None
        Sd = self._dev(S, (n, m)) if m else None
        Yd = self._dev(Y, (n, m)) if m else None
        small = [np.ascontiguousarray(np.asarray(a, dtype=np.float64)).reshape(m, m) if m else None for a in (SS, L, D)]
        dz = torch.empty(N, dtype=torch.float64, device=self.device)
        st = LbfgsStats()
        hp = lambda a: c_void_p(0) if a is None else a.ctypes.data_as(c_void_p)     # noqa: E731
        self._ck(self.lib.pyipm_lbfgs_set_stream(self.h, c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        self._ck(self.lib.pyipm_lbfgs_direction(self.h, self._ptr(g), self._ptr(s), self._ptr(lda), float(zeta), m,
                                                self._ptr(Sd), max(m, 1), self._ptr(Yd), max(m, 1), hp(small[0]),
                                                hp(small[1]), hp(small[2]), float(reg), float(eps), self._ptr(dz),
                                                1 if flip else 0, MEM_DEVICE, ctypes.byref(st)))
        return dz, st.as_dict()

    def last_timings(self):
        out = (c_double * 8)()
        self._ck(self.lib.pyipm_lbfgs_last_timings(self.h, out))
        keys = ("total_ms", "gram_ms", "factor_ms", "solves_ms", "jacobian_passes_ms", "small_ms", "gram_flops",
                "gram_launches")
        return {k: out[i] for i, k in enumerate(keys)}


__all__ = ["LbfgsCore", "LbfgsStats", "MEM_DEVICE", "MEM_HOST", "exported_symbols"]
