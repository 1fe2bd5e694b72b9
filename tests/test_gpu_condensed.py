"""GPU parity of the condensed KKT option (set_option("condensed", 1)).

The library eliminates s and lambda_i, factors  [[H + dI + Ji Sigma Ji', Je], [Je', -dc I]]  (dimension
n + me) and expands the solution back to the reference's full (dx, ds, dle, dli).  Mathematically the
same direction as the full system (pyipm.py:816-844 + :1720), so the bars are the full path's: dz <= 1e-10
relative against the oracle's LU of the FULL matrix, inertia reported for the FULL matrix."""
import os

import numpy as np
import pytest

from oracle import newton_oracle as orc
from pyipm_amd.problems import example_problem, unit_test_x0, make_qp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_DZ = 1e-10


def _core(n, me, mi, condensed=True, **kw):
    from pyipm_amd.newton import NewtonCore
    core = NewtonCore(n, me, mi, device=0, **kw)
    if condensed:
        core.set_option("condensed", 1)
    return core


def _stage(core, qp):
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def _condensed_matrix(qp, n, me, mi, delta=0.0, delta_c=0.0):
    """numpy restatement of the condensed matrix from the reference's blocks (Sigma: pyipm.py:498)."""
    eps = np.finfo(np.float64).eps
    H = np.triu(qp["d2L"]) + np.triu(qp["d2L"], 1).T + delta * np.eye(n)
    if mi:
        sig = qp["lam"][me:] / (qp["s"] + eps)
        H = H + (qp["Ji"] * sig) @ qp["Ji"].T
    M = np.zeros((n + me, n + me))
    M[:n, :n] = H
    if me:
        M[:n, n:] = qp["Je"]
        M[n:, :n] = qp["Je"].T
        M[n:, n:] = -delta_c * np.eye(me)
    return M


@pytest.mark.parametrize("shape", [(24, 8, 16, 0), (40, 0, 12, 1), (96, 32, 48, 4), (160, 40, 100, 5), (256, 64, 96, 6)])
def test_golden_qp_condensed(shape):
    """The committed QP fixtures (values produced by the unmodified reference): same dz, same inertia."""
    n, me, mi, seed = shape
    d = np.load(os.path.join(GOLD, "qp_n%d_me%d_mi%d_s%d.npz" % shape))
    qp = make_qp(n, me, mi, seed)
    core = _core(n, me, mi)
    _stage(core, qp)
    g = core.residual().cpu().numpy()
    core.assemble(0.0, 0.0)
    # storage now holds the condensed matrix (lower triangle referenced -> triu of the row-major view)
    S = core.kkt_storage().cpu().numpy()
    M = _condensed_matrix(qp, n, me, mi)
    Nc = n + me
    assert S.shape[0] == S.shape[1] == ((Nc + 127) // 128) * 128
    np.testing.assert_allclose(np.triu(S[:Nc, :Nc]), np.triu(M), rtol=0, atol=1e-13 * np.abs(M).max())
    assert np.array_equal(np.triu(S[Nc:, Nc:]), np.eye(S.shape[0] - Nc))
    st = core.factor()
    N = core.N
    assert st["n_neg"] == me + mi == int(d["neg"]) and st["n_zero"] == 0 and st["n_pos"] == N - me - mi
    dz = core.solve(flip=True).cpu().numpy()
    assert relerr(dz, d["dz"]) <= TOL_DZ
    raw = core.solve(flip=False).cpu().numpy()
    r = core.matvec(raw).cpu().numpy() - g                 # backward error against the FULL blocks
    assert np.linalg.norm(r) <= 1e-12 * np.linalg.norm(g)


@pytest.mark.parametrize("shape,nb", [((300, 100, 150, 7), 128), ((700, 200, 300, 8), 256), ((1024, 0, 512, 9), 256),
                                      ((1500, 300, 500, 11), 512), ((640, 64, 1999, 12), 256)])
def test_condensed_vs_oracle_and_full(shape, nb):
    n, me, mi, seed = shape
    qp = make_qp(n, me, mi, seed)
    ref, _, Hc, g = orc.newton_step(qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"],
                                    qp["mu"], n, me, mi, regularise=False)
    core = _core(n, me, mi, nb=nb)
    _stage(core, qp)
    dz, st = core.step(0.0, 0.0)
    assert st["n_neg"] == me + mi and st["n_zero"] == 0 and st["n_pos"] == core.N - me - mi
    assert relerr(dz.cpu().numpy(), ref) <= TOL_DZ
    assert relerr(core.solve(flip=True, refine=1).cpu().numpy(), ref) <= TOL_DZ
    full = _core(n, me, mi, condensed=False, nb=nb)
    _stage(full, qp)
    dzf, stf = full.step(0.0, 0.0)
    assert (st["n_neg"], st["n_zero"], st["n_pos"]) == (stf["n_neg"], stf["n_zero"], stf["n_pos"])
    assert relerr(dz.cpu().numpy(), dzf.cpu().numpy()) <= TOL_DZ


def _ragged_condensed(count, seed):
    rng = np.random.default_rng(seed)
    edge = [1, 2, 3, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383]
    out = []
    for i in range(count):
        pick = lambda top: int(rng.choice(edge)) if rng.random() < 0.5 else int(rng.integers(1, top))   # noqa: E731
        n = pick(420)
        me = 0 if rng.random() < 0.3 else min(pick(200), n)
        mi = pick(300)                                    # (the option does nothing without inequalities)
        out.append((n, me, mi, 4000 + i, int(rng.choice([128, 256]))))
    return out


# PYIPM_RAGGED_COUNT / PYIPM_RAGGED_SEED: a longer sweep by hand
@pytest.mark.parametrize("n,me,mi,seed,nb", _ragged_condensed(int(os.environ.get("PYIPM_RAGGED_COUNT", "24")),
                                                              int(os.environ.get("PYIPM_RAGGED_SEED", "2025"))))
def test_condensed_ragged_shapes_vs_oracle(n, me, mi, seed, nb):
    """The condensed option on seeded ragged shapes (block boundaries on and off the 16-column K grid of the Gram launch, the 64 / 128
    tile grids and the panels; n + me below one tile; more inequalities than variables) against the oracle's LU of the FULL system."""
    qp = make_qp(n, me, mi, seed)
    ref, _, Hc, g = orc.newton_step(qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"],
                                    qp["mu"], n, me, mi, regularise=False)
    core = _core(n, me, mi, nb=nb)
    _stage(core, qp)
    dz, st = core.step(0.0, 0.0)
    assert st["n_neg"] == me + mi and st["n_zero"] == 0 and st["n_pos"] == core.N - me - mi
    assert relerr(dz.cpu().numpy(), ref) <= max(TOL_DZ, 20 * np.linalg.cond(Hc) * np.finfo(float).eps)
    core.close()


def test_condensed_call_orders_and_toggle():
    """residual/assemble/factor/solve in the reference's order (fused forward), factor-before-residual,
    explicit right-hand sides, and switching the option off again on the same handle."""
    import torch
    n, me, mi = 700, 150, 420
    qp = make_qp(n, me, mi, seed=33)
    ref, _, Hc, g = orc.newton_step(qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"],
                                    qp["mu"], n, me, mi, regularise=False)
    core = _core(n, me, mi)
    _stage(core, qp)
    core.residual(); core.assemble(0.0, 0.0); core.factor()
    a = core.solve(flip=True).cpu().numpy()
    core.assemble(0.0, 0.0); core.factor(); core.residual()
    b = core.solve(flip=True).cpu().numpy()
    assert relerr(a, ref) <= TOL_DZ and relerr(b, ref) <= TOL_DZ
    core.set_option("fuse_forward", 0)
    core.residual(); core.assemble(0.0, 0.0); core.factor()
    c = core.solve(flip=True).cpu().numpy()
    assert relerr(c, ref) <= TOL_DZ
    core.set_option("fuse_forward", 1)
    rng = np.random.default_rng(5)
    rhs = rng.standard_normal(core.N)
    x = core.solve(rhs=torch.from_numpy(rhs).cuda(), flip=False).cpu().numpy()
    assert relerr(Hc @ x, rhs) <= 1e-10
    assert relerr(x, np.linalg.solve(Hc, rhs)) <= 1e-9
    core.set_option("condensed", 0)
    dz, st = core.step(0.0, 0.0)
    assert core.kkt_storage().shape[0] == ((core.N + 127) // 128) * 128
    assert relerr(dz.cpu().numpy(), ref) <= TOL_DZ and st["n_neg"] == me + mi


def test_condensed_inertia_nonconvex_and_rank_deficient():
    """The two reghess branches on the condensed factor: the golden non-convex step (delta loop) and the
    rank-deficient Jacobian (delta_c) report the FULL matrix's inertia, before and after the shifts."""
    d = np.load(os.path.join(GOLD, "step_nonconvex_delta_loop.npz"))
    n, me, mi = int(d["nvar"]), int(d["neq"]), int(d["nineq"])
    core = _core(n, me, mi)
    core.stage_blocks(d["Q"], d["A"].T.copy(), d["G"].T.copy())
    core.stage_vectors(d["Q"] @ d["x"] + d["c"], d["A"] @ d["x"] - 0.1, d["G"] @ d["x"] + 1.0, d["s"], d["lda"],
                       mu=float(d["mu"]))
    core.residual()
    core.assemble(0.0, 0.0)
    st0 = core.factor()
    w = np.linalg.eigvalsh(d["H"])
    assert st0["n_neg"] == int(np.sum(w < 0)) != me + mi
    core.assemble(float(d["delta_out"]), 0.0)
    st1 = core.factor()
    assert st1["n_neg"] == me + mi and st1["n_zero"] == 0
    assert relerr(core.solve(flip=True).cpu().numpy(), d["dz"]) <= TOL_DZ

    d = np.load(os.path.join(GOLD, "step_rankdef_delta_c.npz"))
    n, me, mi = int(d["nvar"]), int(d["neq"]), int(d["nineq"])
    core = _core(n, me, mi)
    core.stage_blocks(d["Q"], d["A"].T.copy(), d["G"].T.copy())
    core.stage_vectors(d["Q"] @ d["x"] + d["c"], d["A"] @ d["x"] - 0.1, d["G"] @ d["x"] + 1.0, d["s"], d["lda"],
                       mu=float(d["mu"]))
    core.residual()
    core.assemble(0.0, 0.0)
    st = core.factor()
    assert st["n_zero"] >= 1 or st["n_neg"] != me + mi or st["d_min"] / st["d_max"] <= np.finfo(float).eps
    eps = np.finfo(float).eps
    delta_c = np.sqrt(eps) * 1e-4 * float(d["mu_host"]) ** 0.4
    core.assemble(float(d["delta_out"]), delta_c)
    st = core.factor()
    assert st["n_neg"] == me + mi and st["n_zero"] == 0
    # cond ~ 1/delta_c: the multipliers of the duplicated rows are only determined as a sum, so compare the
    # primal part with the reference and check the whole vector through the backward error
    dz = core.solve(flip=True).cpu().numpy()
    assert relerr(dz[: n + mi], d["dz"][: n + mi]) <= 1e-6
    g = core.residual()
    raw = core.solve(flip=False)
    assert float((core.matvec(raw) - g).norm() / g.norm()) <= 1e-4       # pivot ~ delta_c ~ 1e-12: growth ~ 1/delta_c
    raw = core.solve(flip=False, refine=2)                                 # refinement against the full blocks repairs it
    assert float((core.matvec(raw) - g).norm() / g.norm()) <= 1e-10


@pytest.mark.parametrize("k", [3, 5, 7, 8, 9, 10])
def test_ipm_condensed_matches_reference_run(k):
    """Whole solves with the condensed backend land on the unmodified reference's answers."""
    from pyipm_amd.ipm import IPM
    d = np.load(os.path.join(GOLD, "trace_p%02d.npz" % k))
    p = example_problem(k)
    ipm = IPM(x0=unit_test_x0()[k], f=p["f"], df=p["df"], d2f=p["d2f"], ce=p["ce"], dce=p["dce"], d2ce=p["d2ce"],
              ci=p["ci"], dci=p["dci"], d2ci=p["d2ci"], Ftol=1.0e-8, verbosity=-1, condensed=True)
    x, s, lda, fval, kkt = ipm.solve()
    assert min(np.linalg.norm(x - gt) for gt in p["ground_truth"]) <= 1e-3
    assert ipm.signal == int(d["signal"])
    assert ipm.iter_count == int(d["n_iter"])
    np.testing.assert_allclose(x, d["x"], rtol=1e-6, atol=1e-8)
    assert np.isclose(float(fval), float(d["fval"]), rtol=1e-7, atol=1e-10)


def test_condensed_large_properties():
    """N = 8192 full / 5120 condensed: backward error via the block mat-vec, linearity, determinism."""
    import torch
    n, me, mi = 4096, 1024, 1536
    qp = make_qp(n, me, mi, seed=21)
    core = _core(n, me, mi)
    _stage(core, qp)
    g = core.residual()
    core.assemble(0.0, 0.0)
    st = core.factor()
    assert st["n_neg"] == me + mi and st["n_zero"] == 0 and st["n_pos"] == core.N - me - mi
    x = core.solve(flip=False)
    assert float((core.matvec(x) - g).norm() / g.norm()) <= 1e-12
    gen = torch.Generator(device="cpu").manual_seed(1)
    b2 = torch.randn(core.N, dtype=torch.float64, generator=gen).cuda()
    x2 = core.solve(rhs=b2, flip=False)
    x3 = core.solve(rhs=2.5 * g + b2, flip=False)
    assert float((x3 - (2.5 * x + x2)).norm() / x3.norm()) <= 1e-11
    core.assemble(0.0, 0.0)
    core.factor()
    assert torch.equal(core.solve(flip=False), x)
    full = _core(n, me, mi, condensed=False)
    _stage(full, qp)
    dzf, _ = full.step(0.0, 0.0)
    dzc, _ = core.step(0.0, 0.0)
    assert float((dzc - dzf).norm() / dzf.norm()) <= TOL_DZ


@pytest.mark.parametrize("smax", [1e6, 1e9, 1e11])
def test_ill_conditioned_tiles_keep_inertia(smax):
    """Dense ill-conditioned diagonal tiles (H + Ji Sigma Ji' with Sigma up to smax): the block pivots are
    explicit inverses, so without the refinement of the block solves (block_refine) the Schur complement
    loses ~eps*cond and the inertia goes wrong near Sigma ~ 1e9-1e10; with it the factor reports the inertia of
    the full matrix and a refined solve reaches a small backward error."""
    n, me, mi = 200, 60, 120
    qp = make_qp(n, me, mi, 4)
    rng = np.random.default_rng(0)
    sig = np.exp(rng.uniform(np.log(1e-12), np.log(smax), mi))
    sig[:5] = smax
    s = rng.uniform(0.5, 2.0, mi)
    lam = np.concatenate([qp["lam"][:me], sig * s])          # Sigma = lam_i / s
    core = _core(n, me, mi)
    core.set_option("condensed_sigma_max", 1e300)            # fold EVERY inequality into the x-x block
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], s, lam, mu=qp["mu"])
    g = core.residual()
    core.assemble(0.0, 0.0)
    st = core.factor()
    assert st["n_neg"] == me + mi and st["n_zero"] == 0, st
    raw = core.solve(flip=False, refine=2)
    assert float((core.matvec(raw) - g).norm() / g.norm()) <= 1e-9
    if smax >= 1e11:
        core.set_option("block_refine", 0)
        core.assemble(0.0, 0.0)
        assert core.factor()["n_neg"] != me + mi              # the unrefined block algorithm does lose it


@pytest.mark.parametrize("theta", [0.0, 0.9, 1.3, 1e4])
def test_adaptive_split_matches_oracle(theta):
    """condensed_sigma_max splits the inequalities: Sigma <= theta eliminated into the x-x block, the others kept
    as rows with -1/Sigma on the diagonal.  Any split solves the same system: theta = 0 keeps every
    inequality row (dimension n+me+mi), a huge theta is the fully condensed form; in between a mix."""
    n, me, mi, seed = 700, 200, 300, 8
    qp = make_qp(n, me, mi, seed)
    ref, _, Hc, g = orc.newton_step(qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"],
                                    qp["mu"], n, me, mi, regularise=False)
    sig = qp["lam"][me:] / qp["s"]
    na = int((sig > theta).sum())
    core = _core(n, me, mi)
    core.set_option("condensed_sigma_max", theta)
    _stage(core, qp)
    core.residual()
    core.assemble(0.0, 0.0)
    S = core.kkt_storage()
    assert S.shape[0] == ((n + me + na + 127) // 128) * 128          # the system really has n + me + |A| rows
    st = core.factor()
    assert st["n_neg"] == me + mi and st["n_zero"] == 0 and st["n_pos"] == core.N - me - mi
    dz = core.solve(flip=True).cpu().numpy()
    assert relerr(dz, ref) <= TOL_DZ
    dz2, st2 = core.step(0.0, 0.0)
    assert relerr(dz2.cpu().numpy(), ref) <= TOL_DZ and st2["n_neg"] == me + mi
    if 0 < na < mi:
        assert 0.9 <= theta <= 1.3


def test_adaptive_split_survives_huge_sigma():
    """Sigma spanning 1e-12 .. 1e14: the large ones stay rows, so the factor keeps the right inertia and the
    direction reaches a small backward error without the refinement the fully condensed form needs."""
    n, me, mi = 200, 60, 120
    qp = make_qp(n, me, mi, 4)
    rng = np.random.default_rng(0)
    sig = np.exp(rng.uniform(np.log(1e-12), np.log(1e14), mi))
    s = rng.uniform(0.5, 2.0, mi)
    lam = np.concatenate([qp["lam"][:me], sig * s])
    core = _core(n, me, mi)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], s, lam, mu=qp["mu"])
    g = core.residual()
    core.assemble(0.0, 0.0)
    st = core.factor()
    assert st["n_neg"] == me + mi and st["n_zero"] == 0
    raw = core.solve(flip=False, refine=1)
    assert float((core.matvec(raw) - g).norm() / g.norm()) <= 1e-10
