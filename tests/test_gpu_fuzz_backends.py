"""The host IPM loop with the HIP Newton backend against the same loop with the oracle backend (tests/backends.py: the reference's
eigenvalue-based reghess and LU, restated) on seeded random NONCONVEX QPs over a box: the inertia-correcting shifts delta of
pyipm.py:1373-1406 are taken on every such solve, and the two runs must agree in signal, iteration count, the shift they end with and
the point they reach.  PYIPM_BACKEND_FUZZ=count[,seed] adds more problems by hand."""
import os

import numpy as np
import pytest

from backends import OracleBackend

pytestmark = pytest.mark.gpu


def _problem(n, me, seed):
    rng = np.random.default_rng(seed)
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    w = rng.uniform(-1.0, 2.0, n)                       # about a third of the curvatures negative
    Q = (V * w) @ V.T
    Q = 0.5 * (Q + Q.T)
    c = rng.standard_normal(n)
    A = rng.standard_normal((me, n)) / np.sqrt(n)
    G = np.vstack([np.eye(n), -np.eye(n)])
    h = -np.ones(2 * n)
    AT, GT = np.ascontiguousarray(A.T), np.ascontiguousarray(G.T)
    Z = np.zeros((n, n))
    prob = dict(f=lambda x: float(0.5 * x @ Q @ x + c @ x), df=lambda x: Q @ x + c, d2f=lambda x: Q,
                ci=lambda x: G @ x - h, dci=lambda x: GT, d2ci=lambda x, lda: Z)
    if me:
        prob.update(ce=lambda x: A @ x, dce=lambda x: AT, d2ce=lambda x, lda: Z)
    return prob


def _cases():
    out = [(6, 0, 1), (40, 5, 2), (70, 0, 3), (130, 20, 4)]
    spec = os.environ.get("PYIPM_BACKEND_FUZZ", "")
    if spec:
        count, seed = (spec.split(",") + ["9"])[:2]
        rng = np.random.default_rng(int(seed))
        for i in range(int(count)):
            n = int(rng.integers(3, 150))
            out.append((n, 0 if rng.random() < 0.4 else int(rng.integers(1, max(2, n // 3))), 2000 + i))
    return out


@pytest.mark.parametrize("n,me,seed", _cases())
def test_hip_backend_tracks_the_oracle_backend_through_inertia_corrections(n, me, seed):
    from pyipm_amd.ipm import IPM
    prob = _problem(n, me, seed)
    kw = dict(x0=np.zeros(n), verbosity=-1, Ktol=1e-6, niter=30, miter=30)
    hip = IPM(**kw, **prob)
    xh, sh, lh, fh, _ = hip.solve()
    ora = IPM(backend=OracleBackend(n, me, 2 * n), **kw, **prob)
    xo, so, lo, fo, _ = ora.solve()
    if n >= 20:
        assert hip.backend.n_factor > hip.iter_count or hip.delta > 0.0      # shifts were taken (a tiny Q can come out convex)
    assert hip.signal == ora.signal
    assert abs(hip.iter_count - ora.iter_count) <= 1
    assert np.isclose(hip.delta, ora.delta, rtol=1e-9, atol=0.0)
    np.testing.assert_allclose(xh, xo, rtol=1e-6, atol=1e-7)
    assert abs(float(fh) - float(fo)) <= 1e-8 * max(1.0, abs(float(fo)))


def _rank_deficient(n, me, seed, reduced=False):
    """Convex QP over a box whose equality constraints repeat a row (and add a combination of two others): Je has no full column
    rank, the reference's rcond test fires and reghess regularises the multiplier block (delta_c, pyipm.py:1383-1389).
    reduced: the same problem with the dependent rows left out."""
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n))
    Q = M @ M.T / n + np.eye(n)
    c = rng.standard_normal(n)
    A = rng.standard_normal((me, n)) / np.sqrt(n)
    A[-1] = A[0]
    if me >= 5:
        A[-2] = 0.5 * A[1] - 2.0 * A[2]
    if reduced:
        A = A[:-2] if me >= 5 else A[:-1]
    G = np.vstack([np.eye(n), -np.eye(n)])
    h = -np.ones(2 * n)
    AT, GT = np.ascontiguousarray(A.T), np.ascontiguousarray(G.T)
    Z = np.zeros((n, n))
    return dict(f=lambda x: float(0.5 * x @ Q @ x + c @ x), df=lambda x: Q @ x + c, d2f=lambda x: Q,
                ce=lambda x: A @ x, dce=lambda x: AT, d2ce=lambda x, lda: Z,
                ci=lambda x: G @ x - h, dci=lambda x: GT, d2ci=lambda x, lda: Z), A.shape[0]


def _rd_cases():
    out = [(8, 2, 1), (50, 6, 2), (100, 12, 3)]
    spec = os.environ.get("PYIPM_BACKEND_FUZZ", "")
    if spec:
        count, seed = (spec.split(",") + ["9"])[:2]
        rng = np.random.default_rng(int(seed) + 1)
        for i in range(int(count)):
            n = int(rng.integers(6, 140))
            out.append((n, int(rng.integers(2, max(3, n // 4))), 3000 + i))
    return out


@pytest.mark.parametrize("n,me,seed", _rd_cases())
def test_dependent_equalities_reach_the_point_of_the_reduced_problem(n, me, seed):
    """Exactly dependent equality constraints.  The reference's own path is not a yardstick here: its LU of the delta_c-regularised
    matrix (rcond ~ 1e-23) returns rounding noise along the non-unique multipliers and SciPy raises LinAlgError ("Matrix is
    singular") on some iterate of most of these problems -- the oracle backend does exactly that.  The HIP backend (shift branch +
    static pivots + refinement) must converge, and to the point the ORACLE backend finds for the same problem with the dependent
    rows left out."""
    from pyipm_amd.ipm import IPM
    prob, _ = _rank_deficient(n, me, seed)
    kw = dict(x0=np.zeros(n), verbosity=-1, Ktol=1e-6, niter=30, miter=30)
    hip = IPM(**kw, **prob)
    xh, sh, lh, fh, _ = hip.solve()
    red, me_r = _rank_deficient(n, me, seed, reduced=True)
    ora = IPM(backend=OracleBackend(n, me_r, 2 * n), **kw, **red)
    xo, so, lo, fo, _ = ora.solve()
    assert hip.signal == ora.signal == 1
    # both runs stop at |KKT| <= 1e-6, at different iterates: f within 1e-6, and x within what strong convexity (Q >= I) then allows
    assert abs(float(fh) - float(fo)) <= 1e-6 * max(1.0, abs(float(fo)))
    np.testing.assert_allclose(xh, xo, rtol=0, atol=2e-3)
    assert np.abs(prob["ce"](xh)).max() <= 1e-6 and prob["ci"](xh).min() >= -1e-8


def _psd_problem(n, me, seed):
    """Convex QP over a box whose Hessian has rank n // 3 (and a block of exactly zero rows): the x-x block of every KKT system is
    singular, the system as a whole is not -- tiles that cannot pivot on their own (static pivots, 2x2 pivots) on every solve."""
    rng = np.random.default_rng(seed)
    k = max(1, n // 3)
    M = rng.standard_normal((n, k))
    M[: n // 4] = 0.0
    Q = M @ M.T / k
    c = rng.standard_normal(n)
    A = rng.standard_normal((me, n)) / np.sqrt(n)
    G = np.vstack([np.eye(n), -np.eye(n)])
    h = -np.ones(2 * n)
    AT, GT = np.ascontiguousarray(A.T), np.ascontiguousarray(G.T)
    Z = np.zeros((n, n))
    prob = dict(f=lambda x: float(0.5 * x @ Q @ x + c @ x), df=lambda x: Q @ x + c, d2f=lambda x: Q,
                ci=lambda x: G @ x - h, dci=lambda x: GT, d2ci=lambda x, lda: Z)
    if me:
        prob.update(ce=lambda x: A @ x, dce=lambda x: AT, d2ce=lambda x, lda: Z)
    return prob


@pytest.mark.parametrize("n,me,seed", [(n, me, s + 100) for n, me, s in _cases()])
def test_hip_backend_tracks_the_oracle_backend_on_rank_deficient_hessians(n, me, seed):
    from pyipm_amd.ipm import IPM
    prob = _psd_problem(n, me, seed)
    kw = dict(x0=np.zeros(n), verbosity=-1, Ktol=1e-6, niter=30, miter=30)
    hip = IPM(**kw, **prob)
    xh, sh, lh, fh, _ = hip.solve()
    ora = IPM(backend=OracleBackend(n, me, 2 * n), **kw, **prob)
    xo, so, lo, fo, _ = ora.solve()
    assert hip.signal == ora.signal == 1
    assert abs(hip.iter_count - ora.iter_count) <= 1
    assert abs(float(fh) - float(fo)) <= 1e-6 * max(1.0, abs(float(fo)))
    # (the minimiser need not be unique along the null space of Q inside the active face: compare what is determined)
    assert np.abs(prob["ci"](xh)).min() >= -1e-8 and (me == 0 or np.abs(prob["ce"](xh)).max() <= 1e-6)
