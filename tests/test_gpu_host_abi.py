"""The C-ABI used the way INTEGRATION.md's reference-side stub uses it: plain ctypes, NumPy HOST
pointers (PYIPM_MEM_HOST), library-owned workspace (workspace = NULL), no torch tensors at all."""
import ctypes
import os

import numpy as np
import pytest

from oracle import newton_oracle as orc
from pyipm_amd.problems import make_qp

pytestmark = pytest.mark.gpu
HOST = 1


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_host_pointer_step_matches_oracle():
    from pyipm_amd import newton
    lib = newton.load_library()
    n, me, mi = 200, 60, 90
    N = n + 2 * mi + me
    qp = make_qp(n, me, mi, seed=31)
    h = ctypes.c_void_p()
    assert lib.pyipm_newton_create(ctypes.byref(h), n, me, mi, 256, 0, 1, 0, None, 0, None) == 0
    d2L = np.ascontiguousarray(qp["d2L"]); Je = np.ascontiguousarray(qp["Je"]); Ji = np.ascontiguousarray(qp["Ji"])
    # a padded leading dimension on the host side must be honoured
    Jpad = np.zeros((n, mi + 7)); Jpad[:, :mi] = Ji
    assert lib.pyipm_newton_stage_blocks(h, _p(d2L), n, _p(Je), me, _p(Jpad), mi + 7, HOST) == 0
    assert lib.pyipm_newton_stage_vectors(h, _p(qp["df"]), _p(qp["ce"]), _p(qp["ci"]), _p(qp["s"]), _p(qp["lam"]),
                                          0.2, float(np.finfo(float).eps), HOST) == 0
    g = np.empty(N)
    assert lib.pyipm_newton_residual(h, _p(g), HOST) == 0
    assert lib.pyipm_newton_assemble(h, 0.0, 0.0) == 0
    st = newton.FactorStats()
    assert lib.pyipm_newton_factor(h, ctypes.byref(st)) == 0
    dz = np.empty(N)
    assert lib.pyipm_newton_solve(h, None, _p(dz), 1, 0, HOST) == 0
    ref, _, Hc, gref = orc.newton_step(qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"],
                                       qp["mu"], n, me, mi, regularise=False)
    np.testing.assert_allclose(g, gref, rtol=0, atol=1e-13 * np.abs(gref).max())
    assert np.linalg.norm(dz - ref) / np.linalg.norm(ref) <= 1e-10
    assert st.n_neg == me + mi and st.n_zero == 0
    # explicit right-hand side from host memory, no flip; and the block mat-vec
    b = np.random.default_rng(0).standard_normal(N)
    x = np.empty(N)
    assert lib.pyipm_newton_solve(h, _p(b), _p(x), 0, 1, HOST) == 0
    assert np.linalg.norm(Hc @ x - b) <= 1e-12 * np.linalg.norm(b)
    y = np.empty(N)
    assert lib.pyipm_newton_kkt_matvec(h, _p(x), _p(y), HOST) == 0
    np.testing.assert_allclose(y, Hc @ x, rtol=0, atol=1e-12 * np.abs(Hc @ x).max())
    # fused convenience entry point
    dz2 = np.empty(N)
    assert lib.pyipm_newton_step(h, 0.0, 0.0, 0, _p(dz2), ctypes.byref(st), HOST) == 0
    assert np.linalg.norm(dz2 - ref) / np.linalg.norm(ref) <= 1e-10
    assert lib.pyipm_newton_destroy(h) == 0


def test_error_codes_and_call_order():
    from pyipm_amd import newton
    lib = newton.load_library()
    h = ctypes.c_void_p()
    assert lib.pyipm_newton_create(ctypes.byref(h), 0, 0, 0, 256, 0, 1, 0, None, 0, None) == -1      # n must be > 0
    assert lib.pyipm_newton_create(ctypes.byref(h), 8, 0, 0, 96, 0, 1, 0, None, 0, None) == -1       # nb multiple of 128
    assert lib.pyipm_newton_create(ctypes.byref(h), 8, 0, 0, 256, 99, 1, 0, None, 0, None) == -5     # no such device
    assert lib.pyipm_newton_create(ctypes.byref(h), 8, 2, 3, 256, 0, 1, 0, None, 0, None) == 0
    assert lib.pyipm_newton_assemble(h, 0.0, 0.0) == -1                                              # nothing staged yet
    assert b"stage" in lib.pyipm_newton_last_error(h)
    st = newton.FactorStats()
    assert lib.pyipm_newton_factor(h, ctypes.byref(st)) == -1                                        # not assembled
    x = np.zeros(8 + 6 + 2)
    assert lib.pyipm_newton_solve(h, None, _p(x), 1, 0, HOST) == -1                                  # not factored
    assert lib.pyipm_newton_set_option(h, b"no_such_option", 1.0) == -1
    assert lib.pyipm_newton_destroy(h) == 0


def test_nonfinite_is_reported():
    from pyipm_amd import newton
    lib = newton.load_library()
    n = 16
    h = ctypes.c_void_p()
    assert lib.pyipm_newton_create(ctypes.byref(h), n, 0, 0, 256, 0, 1, 0, None, 0, None) == 0
    Q = np.eye(n); Q[3, 5] = np.nan
    df = np.ones(n)
    assert lib.pyipm_newton_stage_blocks(h, _p(Q), n, None, 1, None, 1, HOST) == 0
    assert lib.pyipm_newton_stage_vectors(h, _p(df), None, None, None, None, 0.2, 2.2e-16, HOST) == 0
    assert lib.pyipm_newton_assemble(h, 0.0, 0.0) == 0
    st = newton.FactorStats()
    assert lib.pyipm_newton_factor(h, ctypes.byref(st)) == -4
    assert st.nonfinite != 0
    assert lib.pyipm_newton_destroy(h) == 0


def test_two_handles_on_two_streams_interleaved():
    """Independent handles are independent: two systems stepped on two HIP streams with their calls
    interleaved give the results of running each alone."""
    import torch
    from pyipm_amd.newton import NewtonCore
    from pyipm_amd.problems import make_qp
    shapes = [(300, 100, 150, 7), (500, 0, 260, 9)]
    qps = [make_qp(*sh) for sh in shapes]
    alone = []
    for (n, me, mi, _), qp in zip(shapes, qps):
        c = NewtonCore(n, me, mi, device=0, nb=128)
        c.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]); c.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        alone.append(c.step(0.0, 0.0)[0].clone())
        c.close()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    cores = []
    for st, (n, me, mi, _), qp in zip(streams, shapes, qps):
        with torch.cuda.stream(st):
            c = NewtonCore(n, me, mi, device=0, nb=128)
            c.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]); c.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        cores.append(c)
    for phase in ("residual", "assemble", "factor", "solve"):
        outs = []
        for st, c in zip(streams, cores):
            with torch.cuda.stream(st):
                outs.append(c.assemble(0.0, 0.0) if phase == "assemble" else getattr(c, phase)())
    torch.cuda.synchronize()
    for dz, ref in zip(outs, alone):
        assert torch.equal(dz, ref)
    for c in cores:
        c.close()


def test_lbfgs_host_pointers_match_oracle():
    """include/pyipm_lbfgs.h the way INTEGRATION.md section 2b uses it: NumPy host pointers for everything
    (PYIPM_MEM_HOST), padded leading dimensions, the timings entry point and a user callback that must NOT be called
    when none is installed."""
    from oracle import lbfgs_oracle as lo
    from pyipm_amd import lbfgs
    lib = lbfgs.load()
    n, me, mi, m = 500, 40, 110, 5
    N = n + 2 * mi + me
    qp = make_qp(n, me, mi, seed=77)
    rng = np.random.default_rng(3)
    S = rng.standard_normal((n, m)) / np.sqrt(n)
    Y = 0.7 * S + 0.05 * rng.standard_normal((n, 4)) @ (rng.standard_normal((4, n)) @ S)
    SY = S.T @ Y
    SS, L, D = np.ascontiguousarray(S.T @ S), np.ascontiguousarray(np.tril(SY, -1)), np.ascontiguousarray(np.diag(np.diag(SY)))
    zeta = float(SY[-1, -1] / SS[-1, -1])
    g = rng.standard_normal(N)
    h = ctypes.c_void_p()
    assert lib.pyipm_lbfgs_create(ctypes.byref(h), n, me, mi, m + 2, 0, 0, None) == 0
    Je = np.ascontiguousarray(qp["Je"])
    Jipad = np.zeros((n, mi + 5)); Jipad[:, :mi] = qp["Ji"]              # padded leading dimension
    Spad = np.zeros((n, m + 3)); Spad[:, :m] = S
    Yc = np.ascontiguousarray(Y)
    dz = np.empty(N)
    st = lbfgs.LbfgsStats()
    # direction before staging: reported, not thrown
    rc = lib.pyipm_lbfgs_direction(h, _p(g), _p(qp["s"]), _p(qp["lam"]), zeta, m, _p(Spad), m + 3, _p(Yc), m, _p(SS), _p(L),
                                   _p(D), 1e-12, float(np.finfo(float).eps), _p(dz), 0, HOST, ctypes.byref(st))
    assert rc == -1 and b"stage the Jacobians" in lib.pyipm_lbfgs_last_error(h)
    assert lib.pyipm_lbfgs_stage_jacobian(h, _p(Je), me, _p(Jipad), mi + 5, HOST) == 0
    for flip in (0, 1):
        rc = lib.pyipm_lbfgs_direction(h, _p(g), _p(qp["s"]), _p(qp["lam"]), zeta, m, _p(Spad), m + 3, _p(Yc), m, _p(SS),
                                       _p(L), _p(D), 1e-12, float(np.finfo(float).eps), _p(dz), flip, HOST, ctypes.byref(st))
        assert rc == 0, lib.pyipm_lbfgs_last_error(h)
        ref = lo.direction(g, zeta, S, Y, SS, L, D, Je=qp["Je"], Ji=qp["Ji"], s=qp["s"], lda=qp["lam"], reg=1e-12)
        if flip:
            ref = orc.flip_multipliers(ref, n, mi)
        assert np.linalg.norm(dz - ref) <= 1e-9 * np.linalg.norm(ref)
        assert st.m == m and st.regularised == 0 and st.n_neg == 0 and st.n_zero == 0
    tm = (ctypes.c_double * 8)()
    assert lib.pyipm_lbfgs_last_timings(h, tm) == 0
    assert tm[0] > 0.0 and tm[7] == 1.0                                  # one Gram launch for the two directions
    assert lib.pyipm_lbfgs_set_option(h, b"block_refine", 1.0) == 0
    assert lib.pyipm_lbfgs_set_option(h, b"no_such_option", 1.0) != 0
    assert lib.pyipm_lbfgs_destroy(h) == 0


def test_no_exception_crosses_the_abi():
    """include/pyipm_newton.h: "no exceptions cross it".  Every extern "C" entry is a function-try-block; a
    std::bad_alloc from a host container (here injected where the tile lists are built) comes back as PYIPM_E_NOMEM,
    any other exception as PYIPM_E_HIP, with a message, and the handle stays usable."""
    from pyipm_amd.newton import NewtonCore, NewtonError
    from pyipm_amd.problems import make_qp
    n, me, mi = 700, 100, 300                      # several panels: the bulk launches build tile lists
    qp = make_qp(n, me, mi, seed=5)
    core = NewtonCore(n, me, mi, device=0, nb=128)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    core.set_option("expert", 1)
    core.set_option("tail_group", 2)                             # (as the handles below: the same grouping, the same bits)
    ref, _ = core.step(0.0, 0.0)
    for fault, code in ((1, -3), (2, -2)):
        fresh = NewtonCore(n, me, mi, device=0, nb=128)          # tile lists are cached per handle: use a new one
        fresh.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        fresh.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        fresh.set_option("expert", 1)
        fresh.set_option("tail_group", 2)                        # (groups of 2: a bulk launch -- a tile list -- after every group)
        fresh.set_option("debug_fault", fault)
        with pytest.raises(NewtonError) as ei:
            fresh.step(0.0, 0.0)
        assert ei.value.code == code, (fault, ei.value.code, str(ei.value))
        assert ("bad_alloc" in str(ei.value)) if fault == 1 else ("injected fault" in str(ei.value))
        import torch
        torch.cuda.synchronize()
        dz, st = fresh.step(0.0, 0.0)                            # the hook is one-shot; the handle still works
        assert torch.equal(dz, ref) and st["n_neg"] == me + mi
        fresh.close()
    core.close()
