"""Pivots that cannot stay inside a 64x64 tile: static pivots + refinement (GESP) against the reference.

The reference solves Hc dz = g with LU + partial pivoting over the whole matrix (pyipm.py:18-20, 1720-1721) and
leaves Hc unshifted whenever its eigen-inertia is right (pyipm.py:1381).  tests/golden/pivot_*.npz hold what the
UNMODIFIED reference computes (oracle/make_golden.py --pivot) on systems whose x-x diagonal tiles are singular or
indefinite on their own: LPs (d2L == 0, n >= 128, me + mi >= n), a zero-diagonal Hessian, variables that enter
linearly, and one genuinely singular system.  Bar: the same shift decision and dz <= 1e-10 relative."""
import os

import numpy as np
import pytest

from oracle import newton_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EPS = float(np.finfo(np.float64).eps)


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def _load(name):
    d = np.load(os.path.join(GOLD, "pivot_%s.npz" % name))
    n, me, mi = int(d["nvar"]), int(d["neq"]), int(d["nineq"])
    x, Q, A, G = d["x"], d["Q"], d["A"], d["G"]
    blocks = dict(d2L=Q, Je=np.ascontiguousarray(A.T) if me else None, Ji=np.ascontiguousarray(G.T) if mi else None,
                  df=Q @ x + d["c"], ce=(A @ x - d["b"]) if me else None, ci=(G @ x - d["h"]) if mi else None)
    return d, n, me, mi, blocks


def _direction(be, d, b, delta_in=0.0):
    return be.direction(b["d2L"], b["Je"], b["Ji"], b["df"], b["ce"], b["ci"], d["s"], d["lda"], float(d["mu"]), delta_in,
                        float(d["mu_host"]), 1e-4, 0.4, np.sqrt(EPS), np.sqrt(EPS), EPS)


@pytest.mark.parametrize("name", ["lp", "lp_eq", "zerodiag", "linear_vars"])
def test_reference_direction_without_a_shift(name):
    from pyipm_amd.ipm import HipNewtonBackend
    d, n, me, mi, b = _load(name)
    be = HipNewtonBackend(n, me, mi, device=0)
    dz, delta, st = _direction(be, d, b)
    assert delta == 0.0 == float(d["delta_out"])                   # reghess' decision (pyipm.py:1381): no shift
    assert st["n_neg"] == me + mi == int(d["neg"]) and st["nonfinite"] == 0
    assert st["n_neg"] + st["n_pos"] == be.core.N
    g = be.core.residual().cpu().numpy()
    np.testing.assert_allclose(g, d["g"], rtol=0, atol=1e-13 * np.abs(d["g"]).max())
    assert relerr(dz, d["dz"]) <= 1e-10, (name, relerr(dz, d["dz"]), st, be.last_solve_info)
    if name in ("lp", "lp_eq"):
        assert st["n_zero"] == n and be.n_static == 1              # every x pivot is static: d2L == 0
        assert be.last_solve_info["converged"] and be.last_solve_info["steps"] <= 4
    if name == "linear_vars":
        assert st["n_zero"] >= 64
    if name == "zerodiag":
        assert st["n_2x2"] >= 1                                    # indefinite tiles: 2x2 pivots, direction verified + refined


def test_unrefined_static_factor_is_only_a_preconditioner():
    """What the refinement is for: the factor of the statically pivoted matrix alone is ~sqrt(eps) away."""
    from pyipm_amd.newton import NewtonCore
    d, n, me, mi, b = _load("lp")
    core = NewtonCore(n, me, mi, device=0)
    core.stage_blocks(b["d2L"], b["Je"], b["Ji"])
    core.stage_vectors(b["df"], b["ce"], b["ci"], d["s"], d["lda"], mu=float(d["mu"]))
    core.residual()
    core.assemble(0.0, 0.0)
    st = core.factor()
    assert st["n_zero"] == n and st["nonfinite"] == 0 and np.isfinite(st["growth"])
    plain = core.solve(flip=True, refine=0).cpu().numpy()
    assert np.isfinite(plain).all()
    e0 = relerr(plain, d["dz"])
    ref = core.solve(flip=True, refine=-1).cpu().numpy()
    info = core.solve_info()
    assert info["converged"] and info["backward_error"] <= 1e-14 and info["backward_error0"] > 1e-12
    assert relerr(ref, d["dz"]) <= 1e-10 < e0
    est = core.rcond()
    assert est["w_min"] > 1e3 * est["static_pivot"] and est["rcond"] > 1e-6        # not singular: ~ the reference's 4e-3
    assert 0.2 < est["rcond"] / float(d["rcond"]) < 5.0


def test_singular_system_takes_the_shift_branch():
    """A variable without curvature that no constraint touches: an exactly zero KKT row.  The reference shifts by
    delta0 (pyipm.py:1390-1397); here the zero pivot becomes a static pivot, the smallest eigenvalue of the factored
    matrix sits AT the perturbation level, and the same branch is taken."""
    from pyipm_amd.ipm import HipNewtonBackend
    d, n, me, mi, b = _load("singular")
    be = HipNewtonBackend(n, me, mi, device=0)
    dz, delta, st = _direction(be, d, b)
    assert delta == float(d["delta_out"]) > 0.0
    assert be.last_rcond is not None and be.last_rcond["w_min"] <= 100 * be.last_rcond["static_pivot"]
    assert st["n_neg"] == me + mi
    # delta_c (8e-13 on the lambda_e diagonal, not taken by the reference here: its eigensolver rounds the zero
    # eigenvalue to -3e-15, oracle/make_golden.py) moves dz by ~1e-12
    assert relerr(dz, d["dz"]) <= 1e-9


def test_oracle_agrees_with_the_fixtures():
    for name in ("lp", "lp_eq", "zerodiag", "linear_vars", "singular"):
        d, n, me, mi, b = _load(name)
        dz, delta, _, _ = orc.newton_step(b["d2L"], b["Je"], b["Ji"], b["df"], b["ce"], b["ci"], d["s"], d["lda"],
                                          float(d["mu"]), n, me, mi, delta=0.0, mu_host=float(d["mu_host"]))
        assert delta == float(d["delta_out"]) and relerr(dz, d["dz"]) <= 1e-11


def test_antidiagonal_blocks():
    """[[0, I], [I, 0]] (128 x 128): the leading tile is exactly zero.  Was the documented limitation of round 1
    (rejected pivots, delta shift needed); now factored with static pivots, inertia (64, 64) as the eigenvalues say,
    and solved to LU accuracy."""
    from pyipm_amd.newton import NewtonCore
    k = 64
    M = np.block([[np.zeros((k, k)), np.eye(k)], [np.eye(k), np.zeros((k, k))]])
    rng = np.random.default_rng(0)
    rhs = rng.standard_normal(2 * k)
    core = NewtonCore(2 * k, 0, 0, device=0)
    core.stage_blocks(np.triu(M))
    core.stage_vectors(np.zeros(2 * k))
    core.assemble(0.0, 0.0)
    st = core.factor()
    assert st["n_zero"] == k and st["nonfinite"] == 0
    # no multiplier rows in this handle, so all static pivots are positive and the Schur complement -I/pert carries
    # the negative half of the spectrum
    assert (st["n_neg"], st["n_pos"]) == (k, k)
    x = core.solve(rhs, flip=False, refine=-1).cpu().numpy()
    assert core.solve_info()["converged"]
    assert relerr(x, np.linalg.solve(M, rhs)) <= 1e-12


def _lp(n, extra, seed):
    """min c'x over the box [-1, 1]^n cut by `extra` random half-spaces that keep 0 strictly feasible."""
    rng = np.random.default_rng(seed)
    R = rng.standard_normal((extra, n)) / np.sqrt(n)
    G = np.vstack([np.eye(n), -np.eye(n), R])
    h = np.concatenate([-np.ones(2 * n), -rng.uniform(0.5, 1.5, extra)])
    c = rng.standard_normal(n)
    Z = np.zeros((n, n))
    GT = np.ascontiguousarray(G.T)
    prob = dict(f=lambda x: float(c @ x), df=lambda x: c.copy(), d2f=lambda x: Z,
                ci=lambda x: G @ x - h, dci=lambda x: GT, d2ci=lambda x, lda: Z)
    return prob, c, G, h


def test_ipm_solves_an_lp_with_128_variables():
    """d2L == 0 with n > 64 used to overflow the factor and abort IPM.solve().  The solve now
    converges to the optimum scipy's LP solver finds (late iterations may still shift, as reghess does once the
    active constraints make the matrix numerically singular)."""
    from scipy.optimize import linprog
    from pyipm_amd.ipm import IPM
    n = 128
    prob, c, G, h = _lp(n, 40, 3)
    p = IPM(x0=np.zeros(n), verbosity=-1, Ktol=1e-7, **prob)
    x, s, lda, fval, kkt = p.solve()
    ref = linprog(c, A_ub=-G, b_ub=-h, bounds=[(None, None)] * n, method="highs")
    assert ref.status == 0
    assert abs(fval - ref.fun) <= 1e-5 * max(1.0, abs(ref.fun)), (fval, ref.fun, p.signal)
    assert p.backend.n_static >= 1                    # directions recovered from statically pivoted factors, no overflow
    assert np.all(G @ x - h >= -1e-6)


def _lp_fuzz():
    """PYIPM_LP_FUZZ=count[,seed]: more seeded random LPs, by hand."""
    spec = os.environ.get("PYIPM_LP_FUZZ", "")
    out = [(7, 0, 11), (65, 9, 12), (130, 77, 13)]
    if spec:
        count, seed = (spec.split(",") + ["5"])[:2]
        rng = np.random.default_rng(int(seed))
        out += [(int(rng.integers(2, 220)), int(rng.integers(0, 90)), 1000 + i) for i in range(int(count))]
    return out


@pytest.mark.parametrize("n,extra,seed", _lp_fuzz())
def test_ipm_solves_random_lps(n, extra, seed):
    """LPs of other sizes (one tile, several tiles, tile boundaries; with and without cuts) against scipy's LP solver: every
    Newton system of such a solve has a zero x-x block, i.e. lives on static pivots / 2x2 pivots and refinement."""
    from scipy.optimize import linprog
    from pyipm_amd.ipm import IPM
    prob, c, G, h = _lp(n, extra, seed)
    p = IPM(x0=np.zeros(n), verbosity=-1, Ktol=1e-7, **prob)
    x, s, lda, fval, kkt = p.solve()
    ref = linprog(c, A_ub=-G, b_ub=-h, bounds=[(None, None)] * n, method="highs")
    assert ref.status == 0
    assert abs(fval - ref.fun) <= 1e-5 * max(1.0, abs(ref.fun)), (fval, ref.fun, p.signal)
    assert np.all(G @ x - h >= -1e-6)


def test_batched_handle_static_pivots_stay_finite():
    """The batched kernel shares the tile inversion: LP-shaped members no longer poison their batch."""
    from pyipm_amd.batched import BatchedNewton
    B, n, me, mi = 4, 128, 0, 160
    rng = np.random.default_rng(1)
    H = np.zeros((B, n, n))
    M = rng.standard_normal((n, n))
    H[1] = M @ M.T / n + np.eye(n)                                  # one regular member among LP-shaped ones
    Ji = rng.standard_normal((B, n, mi)) / np.sqrt(n)
    s = rng.uniform(0.5, 2, (B, mi)); lam = rng.uniform(0.5, 2, (B, mi))
    df = rng.standard_normal((B, n)); ci = rng.standard_normal((B, mi))
    bn = BatchedNewton(n, me, mi, B, device=0)
    dz, stats = bn.step_all(H, None, Ji, df, None, ci, s, lam, mu=0.2)
    dz = dz.cpu().numpy()
    assert np.isfinite(dz).all()
    for b in range(B):
        assert stats[b]["nonfinite"] == 0 and stats[b]["n_neg"] == mi
        assert stats[b]["n_zero"] == (0 if b == 1 else n)
    ref, _, _, _ = orc.newton_step(H[1], None, Ji[1], df[1], None, ci[1], s[1], lam[1], 0.2, n, me, mi)
    assert relerr(dz[1], ref) <= 1e-10


def test_unconverged_refinement_takes_the_shift_branch():
    """ADVICE r2 / VERDICT r2 item 4: a refined solve that misses berr_tol must never be returned silently.  Forced
    here on the LP fixture (every x pivot static): with a bar the refinement cannot meet the first direction is sent
    to the delta / delta_c branch reghess takes on rcond <= eps (pyipm.py:1379-1403); with a bar nothing ever meets
    the backend gives up loudly after its shift budget."""
    from pyipm_amd.ipm import HipNewtonBackend
    d, n, me, mi, b = _load("lp")
    be = HipNewtonBackend(n, me, mi, device=0, max_shift_tries=3)
    calls = {"n": 0}
    solve = be._solve

    def first_one_fails(st):
        dz, ok = solve(st)
        calls["n"] += 1
        return dz, ok and calls["n"] > 1

    be._solve = first_one_fails
    dz, delta, st = _direction(be, d, b)
    assert be.n_unconverged == 1 and calls["n"] == 2
    assert delta == pytest.approx(np.sqrt(EPS))                    # delta0: the first shift of the reference's branch
    assert st["n_neg"] == me + mi and np.isfinite(dz).all()
    assert relerr(dz, d["dz"]) <= 1e-5                             # the shifted system's direction: close, not equal

    be2 = HipNewtonBackend(n, me, mi, device=0, max_shift_tries=3)
    be2.berr_tol = 0.0                                             # nothing converges
    be2.berr_fallback = 0.0                                        # ... and nothing is good enough for the last resort
    be2._at_risk = lambda st: True
    with pytest.raises(RuntimeError, match="refined solve did not reach"):
        _direction(be2, d, b)
    assert be2.n_unconverged >= 3

    # with the budget spent, the best direction seen is returned when it is sqrt(eps)-accurate against its own
    # blocks (the reference returns its LU direction whatever its accuracy) -- counted, never silent
    be3 = HipNewtonBackend(n, me, mi, device=0, max_shift_tries=3)
    be3.berr_tol = 0.0
    be3._at_risk = lambda st: True
    dz3, delta3, st3 = _direction(be3, d, b)
    assert be3.n_inexact == 1 and be3.n_unconverged >= 3 and be3.n_inertia_retries == 0
    assert delta3 == 0.0 and relerr(dz3, d["dz"]) <= 1e-10         # the unshifted direction had the smallest backward error

    # the direction returned above is NOT the handle's last solve (that one is the most shifted).  The fraction-to-
    # the-boundary lengths of the device loop must belong to the direction actually returned.
    be4 = HipNewtonBackend(n, me, mi, device=0, max_shift_tries=3, device_step=True)
    be4.berr_tol = 0.0
    be4._at_risk = lambda st: True
    dz4, delta4, _ = _direction(be4, d, b)
    assert be4.n_inexact == 1 and delta4 == 0.0 and be4._dir_override is not None
    tau = 0.995

    def closed_form(v, dv):
        neg = dv < 0
        return min(1.0, float((-tau * v[neg] / dv[neg]).min())) if neg.any() else 1.0

    want = (closed_form(d["s"], dz4[n:n + mi]), closed_form(d["lda"][me:], dz4[n + mi + me:]))
    got = be4.step_lengths(tau)
    assert got == pytest.approx(want, rel=1e-14)
    last = be4.core.step_lengths(tau)                              # ... the handle's own last solve: another direction
    assert last != pytest.approx(want, rel=1e-6)


# ---- static pivots where they hurt -- a whole LP solve of the reference, late iterates included ------
def _lp_trace():
    d = np.load(os.path.join(GOLD, "lptrace_n128_mi320_s31.npz"))
    n, mi = int(d["n"]), int(d["mi"])
    G, h, c = d["G"], d["h"], d["c"]
    return d, n, mi, G, h, c, np.ascontiguousarray(G.T), np.zeros((n, n))


def test_every_direction_of_the_reference_lp_solve():
    """tests/golden/lptrace_*.npz (oracle/make_golden.py --lp-trace): the UNMODIFIED reference solving an LP with 128
    variables and 320 inequalities -- d2L == 0, so all 128 x pivots of the device are static pivots at EVERY one of
    its 18 iterates, the last ones with Sigma spanning 21 decades and rcond(Hc) = 1.4e-12.  Iterate by iterate: the
    same "no shift" decision as reghess (pyipm.py:1381), the reference's inertia, a direction that satisfies the
    blocks to 1e-11, and agreement with the reference's LU direction to what the conditioning of the system allows
    (its own LU carries an error of ~eps/rcond: 1e-10 while rcond >= 1e-6, eps/rcond x 50 beyond)."""
    from pyipm_amd.ipm import HipNewtonBackend
    d, n, mi, G, h, c, GT, Z = _lp_trace()
    be = HipNewtonBackend(n, 0, mi, device=0)
    worst = 0.0
    for it in range(int(d["n_iter"])):
        x, s, lda = d["it_x"][it], d["it_s"][it], d["it_lda"][it]
        dz, delta, st = be.direction(Z, None, GT, c, None, G @ x - h, s, lda, float(d["it_mu"][it]), float(d["it_delta_in"][it]),
                                     float(d["it_mu_host"][it]), 1e-4, 0.4, np.sqrt(EPS), np.sqrt(EPS), EPS)
        assert delta == 0.0 == float(d["it_delta_out"][it]), (it, delta)
        assert st["n_neg"] == mi == int(d["it_neg"][it]) and st["n_zero"] == n and st["nonfinite"] == 0
        g = d["it_g"][it]
        np.testing.assert_allclose(be.core.residual().cpu().numpy(), g, rtol=0, atol=1e-13 * np.abs(g).max())
        info = be.last_solve_info
        assert info["converged"] and info["backward_error"] <= 1e-11, (it, info)
        ref = d["it_dz_raw"][it].copy()
        ref[n + mi:] *= -1.0                                          # pyipm.py:1723-1725
        tol = max(1e-10, 50.0 * EPS / float(d["it_rcond"][it]))
        e = relerr(dz, ref)
        assert e <= tol, (it, e, tol, float(d["it_rcond"][it]), float(d["it_sigma_spread"][it]))
        worst = max(worst, e / tol)
    assert be.n_static == int(d["n_iter"]) and be.n_unconverged == 0


def test_late_lp_iterate_single_step_fixture():
    """pivot_lp_late.npz: the step of that solve with the widest Sigma (3e21) as a single-step fixture in the format of
    the other pivot_* fixtures -- static pivots together with Sigma spanning >= 1e12, which none of them had."""
    from pyipm_amd.ipm import HipNewtonBackend
    d, n, me, mi, b = _load("lp_late")
    assert float(d["sigma_spread"]) >= 1e12 and float(d["delta_out"]) == 0.0
    be = HipNewtonBackend(n, me, mi, device=0)
    dz, delta, st = _direction(be, d, b, float(d["delta_in"]))
    assert delta == 0.0 and st["n_neg"] == mi == int(d["neg"]) and st["n_zero"] == n
    assert be.last_solve_info["converged"] and be.last_solve_info["backward_error"] <= 1e-11
    assert relerr(dz, d["dz"]) <= max(1e-10, 50.0 * EPS / float(d["rcond"]))
    # ... and the direction satisfies the reference's own matrix: Hc dz_raw = g with Hc rebuilt by the oracle
    H = orc.kkt_matrix(b["d2L"], None, b["Ji"], d["s"], d["lda"], n, me, mi)
    raw = dz.copy(); raw[n + mi:] *= -1.0
    assert np.linalg.norm(H @ raw - d["g"]) <= 1e-10 * np.linalg.norm(d["g"])


def test_ipm_retraces_the_reference_lp_solve():
    """The host loop on the HIP backend from the reference's starting point: same number of Newton steps, same signal,
    the iterates of the reference to what the conditioning along the path allows, the same optimum."""
    from pyipm_amd.ipm import IPM
    d, n, mi, G, h, c, GT, Z = _lp_trace()
    prob = dict(f=lambda x: float(c @ x), df=lambda x: c.copy(), d2f=lambda x: Z,
                ci=lambda x: G @ x - h, dci=lambda x: GT, d2ci=lambda x, lda: Z)
    p = IPM(x0=d["x0"].copy(), verbosity=-1, Ktol=float(d["Ktol"]), **prob)
    p.compile(nvar=n)
    seen = []
    direction = p.backend.direction

    def spy(*a, **k):
        seen.append((np.array(a[6]), np.array(a[7])))                # s, lda handed to the step
        return direction(*a, **k)

    p.backend.direction = spy
    x, s, lda, fval, kkt = p.solve()
    assert p.signal == int(d["signal"]) == 1
    assert len(seen) == int(d["n_iter"])
    acc = 1e-9
    for it, (si, li) in enumerate(seen):
        acc = max(acc, 100.0 * EPS / float(d["it_rcond"][max(it - 1, 0)]))    # errors of the previous direction carry over
        np.testing.assert_allclose(si, d["it_s"][it], rtol=0, atol=acc * max(1.0, np.abs(d["it_s"][it]).max()))
        np.testing.assert_allclose(li, d["it_lda"][it], rtol=0, atol=acc * max(1.0, np.abs(d["it_lda"][it]).max()))
    assert abs(fval - float(d["fval"])) <= 1e-7 * max(1.0, abs(float(d["fval"])))
    np.testing.assert_allclose(x, d["x"], rtol=0, atol=1e-5 * max(1.0, np.abs(d["x"]).max()))
    assert p.backend.n_static == int(d["n_iter"]) and p.backend.n_unconverged == 0


