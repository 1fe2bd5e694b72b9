"""CPU oracle for the pyipm Newton-step hot path.  TEST INFRASTRUCTURE ONLY.

This module restates, in plain NumPy/SciPy, what jkaardal/pyipm computes on the
per-inner-iteration Newton step (``/root/reference/pyipm.py:1717-1725``).  It is
the *checker* for the HIP product path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; nothing under
``pyipm_amd/`` does (the product path fails loudly when the HIP library is
missing — there is no CPU fallback).

Pinning status
--------------
The reference's arithmetic for this path lives in third-party Aesara
(``aesara>=2.2.6``, ``/root/reference/setup.py:21``), which is absent from this
image and not installable offline; the reference's own tests pin only the final
``x`` to 1e-3 (``unit_tests.py:51,405-415``).  The oracle is therefore pinned in
two ways (see ``oracle/make_golden.py`` and ``tests/test_oracle_golden.py``):

  1. against per-iteration traces produced by the UNMODIFIED reference
     ``/root/reference/pyipm.py`` executed in the build container through a
     minimal test-only ``aesara`` stand-in (``oracle/aesara_standin``) whose
     ``Solve`` / ``Eigvalsh`` bottom out in the same SciPy routines Aesara's ops
     forward to (``scipy.linalg.solve(assume_a='gen')``,
     ``scipy.linalg.eigvalsh(a, b, lower=True)``);
  2. against the reference's own end-to-end ground truths (p1, p4, p5, p10 of
     ``unit_tests.py`` and the README problem-7 transcript).

Because (1) substitutes the Aesara layer, the honest label is: parity pinned
against the reference's Python executed over a substituted third-party layer;
"parity unpinned" with respect to a real Aesara install.

Block order everywhere: [x (n) | s (mi) | lambda_e (me) | lambda_i (mi)],
N = n + 2*mi + me  (``pyipm.py:824-825``).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg

EPS = float(np.finfo(np.float64).eps)


def kkt_dim(n: int, me: int, mi: int) -> int:
    """N = nvar + 2*nineq + neq  (pyipm.py:824-825)."""
    return n + 2 * mi + me


def kkt_residual(df, Je, Ji, ce, ci, s, lam, mu, n, me, mi, eps=EPS):
    """KKT residual ``self.grad(x, s, lda)`` (NOT negated).

    Follows pyipm.py:655-668 (symbolic) / :610-653 (NumPy twin):
      [ df - Je.lam_e - Ji.lam_i ;  lam_i - mu/(s+eps) ;  ce ;  ci - s ]
    Je is n x me, Ji is n x mi (transposed Jacobians, pyipm.py:486-487,500-501).
    """
    g = np.zeros(kkt_dim(n, me, mi))
    gx = np.array(df, dtype=np.float64).reshape(n).copy()
    if me:
        gx = gx - np.dot(np.asarray(Je).reshape(n, me), lam[:me])
    if mi:
        gx = gx - np.dot(np.asarray(Ji).reshape(n, mi), lam[me:])
    g[:n] = gx
    if mi:
        g[n:n + mi] = lam[me:] - mu / (s + eps)
        g[n + mi + me:] = np.asarray(ci).reshape(mi) - s
    if me:
        g[n + mi:n + mi + me] = np.asarray(ce).reshape(me)
    return g


def kkt_matrix(d2L, Je, Ji, s, lam, n, me, mi, eps=EPS):
    """KKT matrix ``self.hess(x, s, lda)``.

    Follows pyipm.py:816-844: only triu(d2L) is read (:826-827), Je / Ji fill the
    upper-right blocks (:828-835), Sigma = diag(lam_i/(s+eps)) (:498, :836-837),
    -I couples s and lam_i (:838-842), then ``triu(h) + triu(h).T`` minus half the
    doubled diagonal (:843-844).  The NumPy twin (:768-814) builds the same matrix.
    """
    N = kkt_dim(n, me, mi)
    h = np.zeros((N, N))
    h[:n, :n] = np.triu(np.asarray(d2L, dtype=np.float64).reshape(n, n))
    if me:
        h[:n, n + mi:n + mi + me] = np.asarray(Je).reshape(n, me)
    if mi:
        h[:n, n + mi + me:] = np.asarray(Ji).reshape(n, mi)
        h[n:n + mi, n:n + mi] = np.diag(lam[me:] / (s + eps))
        h[n:n + mi, n + mi + me:] = -np.eye(mi)
    h = np.triu(h) + np.triu(h).T
    h = h - np.diag(np.diagonal(h) / 2.0)
    return h


def eigvalsh_ref(M):
    """``self.eigh`` = eigvalsh(M, eye(N)) — generalised form with B = I (pyipm.py:906-909)."""
    return scipy.linalg.eigvalsh(M, np.eye(M.shape[0]), lower=True)


def reghess(Hc, n, me, mi, delta, mu_host, eta=1.0e-4, beta=0.4,
            reg_coef=None, delta0=None, eps=EPS, max_tries=60, stats=None):
    """Inertia / conditioning regulariser ``IPM.reghess`` (pyipm.py:1373-1406).

    Mutates ``Hc`` in place like the reference; returns ``(Hc, delta)`` where
    ``delta`` is the persisted ``self.delta`` (initialised to 0 at pyipm.py:1628).
    ``stats`` (optional dict) receives the number of eigen-decompositions and
    whether the delta_c branch fired.  ``max_tries`` bounds the x10 loop (the
    reference's loop is unbounded).
    """
    if reg_coef is None:
        reg_coef = np.sqrt(eps)          # pyipm.py:353
    if delta0 is None:
        delta0 = reg_coef                # pyipm.py:372
    neigh = 1
    used_dc = False
    w = eigvalsh_ref(Hc)
    rcond = np.min(np.abs(w)) / np.max(np.abs(w))
    if rcond <= eps or (me + mi) != np.sum(w < -eps):
        if rcond <= eps and me:
            i1 = n + mi
            i2 = i1 + me
            Hc[i1:i2, i1:i2] -= reg_coef * eta * (mu_host ** beta) * np.eye(me)
            used_dc = True
        if delta == 0.0:
            delta = delta0
        else:
            delta = np.max([delta / 2, delta0])
        Hc[:n, :n] += delta * np.eye(n)
        w = eigvalsh_ref(Hc)
        neigh += 1
        tries = 0
        while (me + mi) != np.sum(w < -eps):
            Hc[:n, :n] -= delta * np.eye(n)
            delta *= 10.0
            Hc[:n, :n] += delta * np.eye(n)
            w = eigvalsh_ref(Hc)
            neigh += 1
            tries += 1
            if tries >= max_tries:
                raise RuntimeError("reghess: inertia not corrected after %d shifts" % tries)
    if stats is not None:
        stats["n_eigh"] = neigh
        stats["delta_c_used"] = used_dc
        stats["neg"] = int(np.sum(w < -eps))
        stats["rcond"] = float(np.min(np.abs(w)) / np.max(np.abs(w)))
    return Hc, float(delta)


def sym_solve(A, b):
    """``sym_solve`` — a GENERAL LU solve despite the name (pyipm.py:18-20)."""
    return scipy.linalg.solve(A, b, assume_a='gen')


def flip_multipliers(dz, n, mi):
    """``dz[nvar+nineq:] = -dz[nvar+nineq:]`` (pyipm.py:1723-1725)."""
    dz = np.array(dz, dtype=np.float64)
    dz[n + mi:] = -dz[n + mi:]
    return dz


def newton_step(d2L, Je, Ji, df, ce, ci, s, lam, mu, n, me, mi, delta=0.0,
                mu_host=None, eta=1.0e-4, beta=0.4, eps=EPS, regularise=True,
                stats=None):
    """One full reference Newton step (pyipm.py:1717-1725).

    Returns ``(dz, delta, Hc, g)`` with the multiplier sign flip applied to dz.
    With ``regularise=False`` the eigen-inertia test is skipped (the no-retry
    case the CPU baseline also reports separately).
    """
    if mu_host is None:
        mu_host = mu
    g = -kkt_residual(df, Je, Ji, ce, ci, s, lam, mu, n, me, mi, eps)
    Hc = kkt_matrix(d2L, Je, Ji, s, lam, n, me, mi, eps)
    if regularise:
        Hc, delta = reghess(Hc, n, me, mi, delta, mu_host, eta, beta, eps=eps, stats=stats)
    dz = sym_solve(Hc, g.reshape((g.size, 1))).reshape((g.size,))
    if me or mi:
        dz = flip_multipliers(dz, n, mi)
    return dz, delta, Hc, g


def inertia_from_eig(Hc, eps=EPS):
    """(#w < -eps, #|w| <= eps, #w > eps) the way reghess counts (pyipm.py:1381)."""
    w = eigvalsh_ref(Hc)
    return int(np.sum(w < -eps)), int(np.sum(np.abs(w) <= eps)), int(np.sum(w > eps))
