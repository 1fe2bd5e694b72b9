"""GPU parity tests: the HIP Newton-step core (through the C-ABI) against the CPU oracle and the
committed golden vectors.  Bars: KKT assembly bit-exact; residual <= 1e-13 rel (summation order);
search direction dz <= 1e-10 rel fp64 on well-conditioned systems (BASELINE.json north_star);
inertia equal to the eigen-inertia the reference's reghess would compute (pyipm.py:1378-1381)."""
import glob
import os

import numpy as np
import pytest

from oracle import newton_oracle as orc
from pyipm_amd.problems import example_problem, make_qp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_DZ = 1e-10


def _core(n, me, mi, **kw):
    from pyipm_amd.newton import NewtonCore
    return NewtonCore(n, me, mi, device=0, **kw)


def _blocks(prob, x, lda):
    n, me, mi = prob["nvar"], prob["neq"], prob["nineq"]
    d2L = np.array(prob["d2f"](x), dtype=np.float64)
    Je = Ji = ce = ci = None
    if me:
        d2L = d2L - prob["d2ce"](x, lda)
        Je, ce = prob["dce"](x), prob["ce"](x)
    if mi:
        d2L = d2L - prob["d2ci"](x, lda)
        Ji, ci = prob["dci"](x), prob["ci"](x)
    return d2L, Je, Ji, np.asarray(prob["df"](x), dtype=np.float64), ce, ci


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def test_library_loads_on_gpu():
    from pyipm_amd import newton
    assert len(newton.exported_symbols()) >= 25
    peak = newton.mfma_f64_peak(0, 2000)
    assert 10.0 < peak < 200.0, peak


@pytest.mark.parametrize("shape", [(24, 8, 16, 0), (40, 0, 12, 1), (40, 12, 0, 2), (64, 0, 0, 3),
                                   (96, 32, 48, 4), (160, 40, 100, 5), (256, 64, 96, 6)])
def test_qp_golden_step(shape):
    """Golden QP systems: assembly bit-exact vs oracle, g, dz and inertia vs the reference's values."""
    n, me, mi, seed = shape
    d = np.load(os.path.join(GOLD, "qp_n%d_me%d_mi%d_s%d.npz" % shape))
    qp = make_qp(n, me, mi, seed)
    core = _core(n, me, mi)
    N = core.N
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    g = core.residual().cpu().numpy()
    np.testing.assert_allclose(g, d["g"], rtol=0, atol=1e-13 * np.abs(d["g"]).max())
    core.assemble(0.0, 0.0)
    H = orc.kkt_matrix(qp["d2L"], qp["Je"], qp["Ji"], qp["s"], qp["lam"], n, me, mi)
    S = core.kkt_storage().cpu().numpy()
    assert np.array_equal(np.triu(S[:N, :N]), np.triu(H)), "device KKT storage must equal triu(H) bit for bit"
    pad = S[N:, N:]
    assert np.array_equal(np.triu(pad), np.eye(pad.shape[0]))
    assert not np.triu(S[:N, N:]).any()
    st = core.factor()
    assert st["n_neg"] == me + mi == int(d["neg"]) and st["n_zero"] == 0 and st["n_pos"] == N - me - mi
    dz = core.solve(flip=True).cpu().numpy()
    assert relerr(dz, d["dz"]) <= TOL_DZ
    # backward error through the block mat-vec (never touches the factor)
    raw = core.solve(flip=False).cpu().numpy()
    r = core.matvec(raw).cpu().numpy() - g
    assert np.linalg.norm(r) <= 1e-12 * np.linalg.norm(g)
    np.testing.assert_allclose(core.matvec(raw).cpu().numpy(), H @ raw, rtol=0, atol=1e-12 * np.abs(H @ raw).max())


@pytest.mark.parametrize("n,me,mi", [(1, 0, 0), (5, 3, 0), (3, 0, 7), (200, 100, 300), (300, 0, 1500), (1100, 700, 200),
                                     (2049, 1025, 1030), (257, 255, 1023), (1024, 0, 1025)])
def test_block_matvec_and_provider_products_on_ragged_shapes(n, me, mi):
    """`Hc v` from the staged blocks (every block passed over once since round 4: k_symv_tiles for the triangle of d2L, k_jac_tiles
    for a Jacobian block's two products) and the provider products against the oracle's assembled matrix: constraint counts below
    one 256-column window, above one 1024-column segment, larger than n, empty blocks; shifts delta / delta_c included."""
    import torch
    qp = make_qp(n, me, mi, 7 + n % 5)
    core = _core(n, me, mi)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    rng = np.random.default_rng(n + 3 * me + 5 * mi)
    N = core.N
    for delta, delta_c in ((0.0, 0.0), (0.37, 1.0e-3)):
        core.assemble(delta, delta_c)
        H = orc.kkt_matrix(qp["d2L"], qp["Je"], qp["Ji"], qp["s"], qp["lam"], n, me, mi)
        H = H + np.diag(np.r_[np.full(n, delta), np.zeros(mi), np.full(me, -delta_c), np.zeros(mi)])
        v = rng.standard_normal(N)
        y = core.matvec(torch.from_numpy(v).cuda()).cpu().numpy()
        ref = H @ v
        np.testing.assert_allclose(y, ref, rtol=0, atol=1e-13 * (np.abs(H) @ np.abs(v)).max())
    x = rng.standard_normal(n)
    Qx, Ax, Gx = core.block_products(torch.from_numpy(x).cuda())
    Qs = np.triu(qp["d2L"]) + np.triu(qp["d2L"], 1).T
    np.testing.assert_allclose(Qx.cpu().numpy(), Qs @ x, rtol=0, atol=1e-13 * (np.abs(Qs) @ np.abs(x)).max())
    if me:
        np.testing.assert_allclose(Ax.cpu().numpy(), qp["Je"].T @ x, rtol=0, atol=1e-13 * (np.abs(qp["Je"]).T @ np.abs(x)).max())
    if mi:
        np.testing.assert_allclose(Gx.cpu().numpy(), qp["Ji"].T @ x, rtol=0, atol=1e-13 * (np.abs(qp["Ji"]).T @ np.abs(x)).max())
    core.close()


@pytest.mark.parametrize("k", range(1, 11))
def test_reference_traces(k):
    """Every Newton system the unmodified reference met while solving problems 1-10 (seed-42 x0):
    same Hc (incl. reghess shifts) -> same dz, same inertia."""
    d = np.load(os.path.join(GOLD, "trace_p%02d.npz" % k))
    prob = example_problem(k)
    n, me, mi = prob["nvar"], prob["neq"], prob["nineq"]
    core = _core(n, me, mi)
    N = core.N
    for it in range(int(d["n_iter"])):
        x, s, lda = d["it_x"][it], d["it_s"][it], d["it_lda"][it]
        d2L, Je, Ji, df, ce, ci = _blocks(prob, x, lda)
        core.stage_blocks(d2L, Je, Ji)
        core.stage_vectors(df, ce, ci, s, lda, mu=float(d["it_mu"][it]))
        g = core.residual().cpu().numpy()
        np.testing.assert_allclose(g, d["it_g"][it], rtol=0, atol=1e-14 * max(1.0, np.abs(g).max()))
        Hc = d["it_Hc"][it]
        delta = float(d["it_delta_out"][it]) if not np.array_equal(Hc, d["it_H"][it]) else 0.0
        # the delta_c branch never fires on these traces; delta shifts the x block only
        core.assemble(delta, 0.0)
        S = core.kkt_storage().cpu().numpy()
        assert np.array_equal(np.triu(S[:N, :N]), np.triu(Hc))
        st = core.factor()
        assert st["n_neg"] == me + mi and st["n_zero"] == 0
        dz = core.solve(flip=True).cpu().numpy()
        cond = np.linalg.cond(Hc)
        assert relerr(dz, d["it_dz"][it]) <= max(TOL_DZ, 1e-15 * cond), (k, it, cond)


def test_problem7_fixed_point():
    d = np.load(os.path.join(GOLD, "step_p07_fixed.npz"))
    prob = example_problem(7)
    d2L, Je, Ji, df, ce, ci = _blocks(prob, d["x"], d["lda"])
    core = _core(3, 1, 3)
    core.stage_blocks(d2L, Je, Ji)
    core.stage_vectors(df, ce, ci, d["s"], d["lda"], mu=float(d["mu"]))
    dz, st = core.step(0.0, 0.0)
    assert relerr(dz.cpu().numpy(), d["dz"]) <= TOL_DZ
    assert st["n_neg"] == 4 == int(d["neg"]) and st["n_2x2"] >= 1   # zero-diagonal Hessian forces 2x2 pivots


def test_inertia_detects_nonconvexity_and_delta_fixes_it():
    """pyipm.py:1381,1399-1403: wrong inertia at delta=0, right inertia at the reference's delta."""
    d = np.load(os.path.join(GOLD, "step_nonconvex_delta_loop.npz"))
    n, me, mi = int(d["nvar"]), int(d["neq"]), int(d["nineq"])
    core = _core(n, me, mi)
    core.stage_blocks(d["Q"], d["A"].T.copy(), d["G"].T.copy())
    lda = d["lda"]
    df = d["Q"] @ d["x"] + d["c"]
    core.stage_vectors(df, d["A"] @ d["x"] - 0.1, d["G"] @ d["x"] + 1.0, d["s"], lda, mu=float(d["mu"]))
    core.assemble(0.0, 0.0)
    st0 = core.factor()
    w = np.linalg.eigvalsh(d["H"])
    assert st0["n_neg"] == int(np.sum(w < 0)) != me + mi
    core.assemble(float(d["delta_out"]), 0.0)
    st1 = core.factor()
    assert st1["n_neg"] == me + mi
    g = core.residual().cpu().numpy()
    np.testing.assert_allclose(g, d["g"], rtol=0, atol=1e-14 * np.abs(d["g"]).max())
    assert relerr(core.solve(flip=True).cpu().numpy(), d["dz"]) <= TOL_DZ


def test_rank_deficient_reports_zero_pivot_then_delta_c():
    """pyipm.py:1383-1389: singular KKT (duplicated equality row) -> a rejected pivot; with the
    reference's delta / delta_c shifts the inertia is right."""
    d = np.load(os.path.join(GOLD, "step_rankdef_delta_c.npz"))
    n, me, mi = int(d["nvar"]), int(d["neq"]), int(d["nineq"])
    core = _core(n, me, mi)
    core.stage_blocks(d["Q"], d["A"].T.copy(), d["G"].T.copy())
    core.stage_vectors(d["Q"] @ d["x"] + d["c"], d["A"] @ d["x"] - 0.1, d["G"] @ d["x"] + 1.0, d["s"], d["lda"],
                       mu=float(d["mu"]))
    core.assemble(0.0, 0.0)
    st = core.factor()
    assert st["n_zero"] >= 1
    eps = np.finfo(float).eps
    delta_c = np.sqrt(eps) * 1e-4 * float(d["mu_host"]) ** 0.4
    core.assemble(float(d["delta_out"]), delta_c)
    S = core.kkt_storage().cpu().numpy()
    N = core.N
    assert np.array_equal(np.triu(S[:N, :N]), np.triu(d["Hc"]))
    st = core.factor()
    assert st["n_neg"] == me + mi and st["n_zero"] == 0


@pytest.mark.parametrize("shape,nb", [((300, 100, 150, 7), 128), ((700, 200, 300, 8), 256), ((1024, 0, 512, 9), 256),
                                      ((900, 400, 0, 10), 128), ((1500, 300, 500, 11), 512)])
def test_multi_panel_vs_oracle(shape, nb):
    """Sizes that span several panels / MFMA trailing updates, checked against the oracle's LU."""
    n, me, mi, seed = shape
    qp = make_qp(n, me, mi, seed)
    core = _core(n, me, mi, nb=nb)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    dz, st = core.step(0.0, 0.0)
    ref, _, Hc, g = orc.newton_step(qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"],
                                    qp["mu"], n, me, mi, regularise=False)
    assert st["n_neg"] == me + mi and st["n_zero"] == 0
    assert relerr(dz.cpu().numpy(), ref) <= TOL_DZ
    dz1 = core.solve(flip=True, refine=1).cpu().numpy()
    assert relerr(dz1, ref) <= TOL_DZ


def test_large_roundtrip_properties():
    """N = 8192: too big for committed vectors; checked through size-independent properties —
    backward error of the solve via the block mat-vec, linearity, and idempotence of refactoring."""
    import torch
    n, me, mi = 4096, 1024, 1536
    qp = make_qp(n, me, mi, seed=21)
    core = _core(n, me, mi)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    g = core.residual()
    core.assemble(0.0, 0.0)
    st = core.factor()
    assert st["n_neg"] == me + mi and st["n_zero"] == 0 and st["n_pos"] == core.N - me - mi
    x = core.solve(flip=False)
    r = core.matvec(x) - g
    assert float(r.norm() / g.norm()) <= 1e-12
    # linearity: solve(a*b1 + b2) == a*solve(b1) + solve(b2)
    gen = torch.Generator(device="cpu").manual_seed(1)
    b2 = torch.randn(core.N, dtype=torch.float64, generator=gen).cuda()
    x2 = core.solve(rhs=b2, flip=False)
    x3 = core.solve(rhs=2.5 * g + b2, flip=False)
    assert float((x3 - (2.5 * x + x2)).norm() / x3.norm()) <= 1e-11
    # determinism / idempotence: refactor the same system -> bitwise the same direction
    core.assemble(0.0, 0.0)
    core.factor()
    assert torch.equal(core.solve(flip=False), x)
    # sign flip touches exactly the multiplier block (x took its forward pass under the factorisation, panel by panel; a
    # second solve runs it as one launch, another summation order: compare with a solve of the same kind)
    xs = core.solve(flip=False)
    xf = core.solve(flip=True)
    assert torch.equal(xf[: n + mi], xs[: n + mi]) and torch.equal(xf[n + mi:], -xs[n + mi:])
    assert float((xs - x).norm() / x.norm()) <= 1e-12


def test_baseline_config2_exact_vs_oracle():
    """BASELINE.json configs[1]: synthetic convex QP, n=2048 vars, 2048 ineq (KKT dim 6144), fp64, one GPU.
    The oracle's LU at this size takes a couple of seconds; dz must agree to <= 1e-10 relative."""
    n, me, mi = 2048, 0, 2048
    qp = make_qp(n, me, mi, seed=0)
    core = _core(n, me, mi)
    core.stage_blocks(qp["d2L"], None, qp["Ji"])
    core.stage_vectors(qp["df"], None, qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    dz, st = core.step(0.0, 0.0)
    ref, _, Hc, g = orc.newton_step(qp["d2L"], None, qp["Ji"], qp["df"], None, qp["ci"], qp["s"], qp["lam"],
                                    qp["mu"], n, me, mi, regularise=False)
    assert core.N == 6144 and st["n_neg"] == mi and st["n_zero"] == 0
    assert relerr(dz.cpu().numpy(), ref) <= TOL_DZ
    S = core.kkt_storage            # (factor overwrote it) -> re-assemble and compare bit for bit
    core.assemble(0.0, 0.0)
    A = core.kkt_storage().cpu().numpy()
    assert np.array_equal(np.triu(A[:6144, :6144]), np.triu(Hc))


@pytest.mark.parametrize("shape,nb", [((2048, 512, 1536, 3), 256), ((1100, 300, 1700, 4), 128), ((1536, 0, 2048, 5), 256)])
def test_assembly_keeps_only_zeros_nothing_can_fill(shape, nb):
    """K1 does not store again the zeros of the (s,x), (s,s), (lambda_e,s) and (lambda_i,s) blocks that no elimination step
    can fill in (k_assemble, zeros_in_place): after a factorisation the storage must still hold exact zeros there, so the
    next assembly is bit for bit triu(H) again -- also after a factorisation that met NaN, after the condensed system used
    the same storage, and with the option off."""
    n, me, mi, seed = shape
    qp = make_qp(n, me, mi, seed)
    rng = np.random.default_rng(seed)
    core = _core(n, me, mi, nb=nb)
    N = core.N
    core.stage_blocks(qp["d2L"], qp["Je"] if me else None, qp["Ji"])

    def check(s_vec, lam):
        core.stage_vectors(qp["df"], qp["ce"] if me else None, qp["ci"], s_vec, lam, mu=qp["mu"])
        core.assemble(0.0, 0.0)
        H = orc.kkt_matrix(qp["d2L"], qp["Je"], qp["Ji"], s_vec, lam, n, me, mi)
        S = core.kkt_storage().cpu().numpy()              # (handing out the pointer makes every later assembly a full one
        assert np.array_equal(np.triu(S[:N, :N]), np.triu(H))
        if keep[0]:
            core.set_option("keep_zeros", 1)              #  until the option is set again: nothing holds the pointer any more)

    def step(s_vec, lam):
        core.stage_vectors(qp["df"], qp["ce"] if me else None, qp["ci"], s_vec, lam, mu=qp["mu"])
        return core.step(0.0, 0.0)

    keep = [True]
    s0, l0 = qp["s"], qp["lam"]
    step(s0, l0)                                             # full assembly + factorisation
    s1 = s0 * rng.uniform(0.5, 2.0, mi); l1 = l0.copy(); l1[me:] *= rng.uniform(0.5, 2.0, mi)
    step(s1, l1)                                             # assembly with the zeros left in place + factorisation
    check(s0, l0)                                            # ... and again: must be triu(H) bit for bit
    step(s1, l1); step(s0, l0)
    sbad = s0.copy(); sbad[mi // 2] = np.nan
    with pytest.raises(Exception):
        step(sbad, l0)                                       # NaN reaches the factorisation: the zeros are not trusted any more
    check(s1, l1)
    step(s0, l0)
    core.set_option("condensed", 1)
    step(s1, l1)                                             # the condensed system overwrites the storage with another layout
    core.set_option("condensed", 0)
    step(s0, l0)
    check(s1, l1)
    core.set_option("keep_zeros", 0)
    keep[0] = False
    step(s0, l0)
    check(s1, l1)


def test_baseline_config3_properties():
    """BASELINE.json configs[2]: n=16384, 8192 eq + 8192 ineq -> KKT dim N = n + 2*mi + me = 40960
    (the reference formula, pyipm.py:824-825; BASELINE.json's '~49k' is approximate).  13.4 GB of KKT
    storage: no CPU oracle at this size, so parity is through size-independent properties."""
    import torch
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_qp_device
    n, me, mi = 16384, 8192, 8192
    dev = torch.device("cuda", 0)
    qp = make_qp_device(n, me, mi, 3, dev)
    core = _core(n, me, mi)
    assert core.N == 40960
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    g = core.residual()
    # residual against a torch fp64 evaluation of the same formula (pyipm.py:655-668)
    lam = qp["lam"]
    gx = -(qp["df"] - qp["Je"] @ lam[:me] - qp["Ji"] @ lam[me:])
    gs = -(lam[me:] - qp["mu"] / (qp["s"] + np.finfo(float).eps))
    ref_g = torch.cat([gx, gs, -qp["ce"], -(qp["ci"] - qp["s"])])
    assert float((g - ref_g).norm() / ref_g.norm()) <= 1e-13
    core.assemble(0.0, 0.0)
    st = core.factor()
    assert st["n_neg"] == me + mi and st["n_zero"] == 0 and st["n_pos"] == core.N - me - mi and st["nonfinite"] == 0
    x = core.solve(flip=False)
    assert float((core.matvec(x) - g).norm() / g.norm()) <= 1e-11          # backward error via the blocks
    x1 = core.solve(flip=False, refine=1)
    assert float((core.matvec(x1) - g).norm() / g.norm()) <= 1e-12
    assert float((x1 - x).norm() / x.norm()) <= 1e-9
    x2 = core.solve(flip=False)                          # (x above took the forward pass fused under the factorisation: other bits)
    dz = core.solve(flip=True)
    assert torch.equal(dz[: n + mi], x2[: n + mi]) and torch.equal(dz[n + mi:], -x2[n + mi:])
    assert float((x2 - x).norm() / x.norm()) <= 1e-12


def test_fused_forward_call_orders():
    """factor() trails the forward substitution of a pending residual; every call order must give the same dz."""
    n, me, mi = 700, 200, 300
    qp = make_qp(n, me, mi, 8)
    core = _core(n, me, mi)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    # (bit for bit with the per-panel sweep launches, which the fused forward pass uses too; with the one-launch sweeps a
    # forward pass under the factorisation and one after it sum in different orders: compared to rounding at the end)
    core.set_option("sweep_persist", 0)
    dz_step, _ = core.step(0.0, 0.0)                      # fused convenience call
    g = core.residual(); core.assemble(0.0, 0.0); core.factor()
    dz_a = core.solve(flip=True)                          # picks up the fused forward pass
    dz_b = core.solve(flip=True)                          # second solve: full substitution again
    dz_c = core.solve(rhs=g, flip=True)                   # explicit right-hand side
    core.set_option("fuse_forward", 0)
    core.residual(); core.assemble(0.0, 0.0); core.factor()
    dz_d = core.solve(flip=True)                          # no fusion at all
    core.set_option("fuse_forward", 1)
    core.residual(); core.assemble(0.0, 0.0); core.factor()
    y = core.matvec(dz_step)                              # touches the work vectors between factor and solve
    dz_e = core.solve(flip=True)
    import torch
    for other in (dz_a, dz_b, dz_c, dz_d, dz_e):
        assert torch.equal(other, dz_step)
    core.set_option("sweep_persist", 1)
    core.residual(); core.assemble(0.0, 0.0); core.factor()
    for other in (core.solve(flip=True), core.solve(flip=True), core.solve(rhs=g, flip=True)):
        assert float((other - dz_step).norm() / dz_step.norm()) <= 1e-13
    # retry pattern of the host loop: re-assemble with a shift, factor again, solve
    core.residual(); core.assemble(0.0, 0.0); core.factor(); core.assemble(1e-3, 0.0); core.factor()
    dz_shift = core.solve(flip=True)
    H = orc.kkt_matrix(qp["d2L"], qp["Je"], qp["Ji"], qp["s"], qp["lam"], n, me, mi)
    H[:n, :n] += 1e-3 * np.eye(n)
    gg = -orc.kkt_residual(qp["df"], qp["Je"], qp["Ji"], qp["ce"], qp["ci"], qp["s"], qp["lam"], qp["mu"], n, me, mi)
    ref = orc.flip_multipliers(np.linalg.solve(H, gg), n, mi)
    assert relerr(dz_shift.cpu().numpy(), ref) <= TOL_DZ


def _ragged_shapes(count, seed):
    """Seeded ragged shapes: block boundaries on and off the 64 / 128 / nb grid, empty blocks, tiny systems."""
    rng = np.random.default_rng(seed)
    edge = [1, 2, 3, 63, 64, 65, 127, 128, 129, 191, 192, 255, 256, 257, 320, 383, 385]
    out = []
    for i in range(count):
        pick = lambda top: int(rng.choice(edge)) if rng.random() < 0.5 else int(rng.integers(1, top))   # noqa: E731
        n = pick(420)
        me = 0 if rng.random() < 0.25 else min(pick(200), n)          # at most n independent equalities
        mi = 0 if rng.random() < 0.25 else pick(260)
        out.append((n, me, mi, 1000 + i, int(rng.choice([128, 256, 512]))))
    return out


# PYIPM_RAGGED_COUNT / PYIPM_RAGGED_SEED: a longer sweep over other shapes by hand (round 4: 600 shapes of three seeds, no failure)
@pytest.mark.parametrize("n,me,mi,seed,nb", _ragged_shapes(int(os.environ.get("PYIPM_RAGGED_COUNT", "40")),
                                                           int(os.environ.get("PYIPM_RAGGED_SEED", "2024"))))
def test_ragged_shapes_vs_oracle(n, me, mi, seed, nb):
    """Forty seeded ragged (n, me, mi, nb) combinations against the oracle's LU and the eigen-inertia rule: the skipping
    of structurally zero tiles, the closed-form slack panels and the group schedule all depend on where the block
    boundaries fall relative to tiles, panels and groups."""
    qp = make_qp(n, me, mi, seed)
    core = _core(n, me, mi, nb=nb)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    dz, st = core.step(0.0, 0.0)
    ref, _, Hc, g = orc.newton_step(qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"],
                                    qp["mu"], n, me, mi, regularise=False)
    assert st["n_neg"] == me + mi and st["n_zero"] == 0 and st["n_pos"] == n + mi
    cond = np.linalg.cond(Hc)
    assert relerr(dz.cpu().numpy(), ref) <= max(TOL_DZ, 20 * cond * np.finfo(float).eps)
    core.close()


@pytest.mark.parametrize("n,me,mi,nb,tail_cols", [(1500, 700, 1111, 128, 2048), (2049, 0, 1500, 128, 1024), (3000, 1000, 0, 256, 4096),
                                                  (1000, 333, 2500, 128, 0), (2600, 513, 1300, 256, 24576), (4100, 1, 2047, 512, 3000)])
def test_ragged_multi_group_shapes_vs_oracle(n, me, mi, nb, tail_cols):
    """Mid-size ragged systems (N up to 8200: tens of panels, both group sizes of the schedule in one factorisation via
    `tail_cols`) against the oracle's LU; bitwise equal with and without the structural-zero skipping."""
    import torch
    qp = make_qp(n, me, mi, n % 97)
    ref, _, Hc, g = orc.newton_step(qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"],
                                    qp["mu"], n, me, mi, regularise=False)
    out = []
    for skip in (1, 0):
        core = _core(n, me, mi, nb=nb)
        core.set_option("expert", 1)
        core.set_option("tail_group", 2 if tail_cols else 4)        # (group sizes 4 and 2; the column threshold itself is fixed since round 6)
        core.set_option("skip_zeros", skip)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        dz, st = core.step(0.0, 0.0)
        assert st["n_neg"] == me + mi and st["n_zero"] == 0
        out.append(dz.clone())
        core.close()
    assert torch.equal(out[0], out[1])
    assert relerr(out[0].cpu().numpy(), ref) <= 1e-9


def test_exported_storage_pointer_disables_zero_keeping_for_good():
    """a caller that keeps the tensor from kkt_storage() and writes into the structurally zero blocks LATER
    must not corrupt the assemblies that follow: once the pointer is out every assembly stores every entry, until the
    option is set again."""
    n, me, mi, seed = 320, 64, 192, 5
    qp = make_qp(n, me, mi, seed)
    core = _core(n, me, mi, nb=256)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    dz0, _ = core.step(0.0, 0.0)
    dz1, _ = core.step(0.0, 0.0)                             # zeros kept in place
    assert torch_equal(dz0, dz1)
    S = core.kkt_storage()                                   # a view the caller keeps
    core.step(0.0, 0.0)                                      # full assembly (the old behaviour re-armed the shortcut here)
    S[:n, n:n + mi] = 7.0                                    # rows = columns of the lower triangle: the (s,x) block
    dz2, _ = core.step(0.0, 0.0)
    assert torch_equal(dz0, dz2)


def torch_equal(a, b):
    import torch
    return bool(torch.equal(a, b))
