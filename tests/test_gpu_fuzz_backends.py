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
