"""CPU restatement of the reference's L-BFGS search direction.  TEST INFRASTRUCTURE ONLY.

Not part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s CPU-baseline leg import this module
(the product's L-BFGS direction is ``pyipm_lbfgs_direction`` in the HIP library and fails loudly
without it).  Each function names the reference lines it follows; terminal arithmetic is the same
SciPy routine the reference reaches through Aesara (``scipy.linalg.solve(assume_a='gen')`` for
``sym_solve`` pyipm.py:18-20, ``numpy.linalg.eigh`` for ``eigh`` pyipm.py:1108).

Pinned by ``tests/golden/lbfgs_*.npz``: per-iteration (x, s, lda, g, zeta, S, Y, SS, L, D) -> dz
records of the UNMODIFIED reference run with ``lbfgs=4`` (the unit-test setting,
unit_tests.py:49) on its ten example problems, and direction records on synthetic QP-shaped
storage (generator: ``oracle/make_golden.py --lbfgs``).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg


def _solve(a, b):
    return scipy.linalg.solve(a, b, assume_a='gen')


def lbfgs_init(n, zeta0=1.0):
    """pyipm.py:993-1005."""
    e = np.zeros
    return float(zeta0), e((n, 0)), e((n, 0)), e((0, 0)), e((0, 0)), e((0, 0)), 0


def lbfgs_update(x_old, x_new, g_old, g_new, zeta, S, Y, SS, L, D, lbfgs_fail, n, constrained, memory, eps,
                 zeta0=1.0):
    """pyipm.py:1282-1371.  ``memory`` is the constructor's ``lbfgs``; the storage grows to memory+1
    pairs before the oldest is dropped (the reference tests ``S.shape[1] > self.lbfgs``, :1300)."""
    dx = x_new - x_old
    dg = g_old[:n] - g_new[:n]
    if constrained:
        zeta_new = np.dot(dg, dx) / (np.dot(dx, dx) + eps)
    else:
        zeta_new = np.dot(dg, dx) / (np.dot(dg, dg) + eps)
    if np.dot(dx, dg) > np.sqrt(eps) and zeta_new > np.sqrt(eps):
        zeta = zeta_new
        if S.shape[1] > memory:
            S, Y, SS, L, D = S.copy(), Y.copy(), SS.copy(), L.copy(), D.copy()
            S[:, :-1] = S[:, 1:]
            Y[:, :-1] = Y[:, 1:]
            SS[:-1, :-1] = SS[1:, 1:]
            L[:-1, :-1] = L[1:, 1:]
            D[:-1, :-1] = D[1:, 1:]
        else:
            k = S.shape[1] + 1
            grow = lambda Mx: np.pad(Mx, ((0, 1), (0, 1)))      # noqa: E731
            S = np.concatenate([S, np.zeros((n, 1))], axis=1)
            Y = np.concatenate([Y, np.zeros((n, 1))], axis=1)
            SS, L, D = grow(SS), grow(L), grow(D)
            assert SS.shape == (k, k)
        S[:, -1] = dx
        Y[:, -1] = dg
        if constrained:
            upd = S.T @ dx
        else:
            upd = Y.T @ dg                                      # "SS" holds Y'Y for unconstrained problems
        SS[:, -1] = upd
        SS[-1, :] = upd
        if constrained:
            L[-1, :] = dx @ Y
            L[-1, -1] = 0.0
        else:
            L[:, -1] = S.T @ dg                                 # "L" holds R (upper triangular S'Y)
        D[-1, -1] = np.dot(dx, dg)
        lbfgs_fail = 0
    else:
        lbfgs_fail += 1
    if lbfgs_fail > memory and S.shape[1] > 0:                  # lbfgs_fail_max = lbfgs (:360)
        return lbfgs_init(n, zeta0)                             # :1363-1368
    return zeta, S, Y, SS, L, D, lbfgs_fail


def direction_unconstrained(g, zeta, S, Y, SS, L, D):
    """pyipm.py:1149-1175: dz = zeta*g + W Q W' g with W = [S, zeta*Y] (inverse-Hessian form)."""
    n = g.size
    m = S.shape[1]
    Hg = zeta * g.reshape(n, 1)
    if m == 0:
        return Hg.reshape(n)
    W = np.concatenate([S, zeta * Y], axis=1)
    WT_g = W.T @ g
    B = -_solve(L, WT_g[:m].reshape(m, 1))
    A = -_solve(L.T, (D + zeta * SS) @ B) - _solve(L.T, WT_g[m:].reshape(m, 1))
    return (Hg + W @ np.concatenate([A, B], axis=0)).reshape(n)


def direction_constrained(g, zeta, S, Y, SS, L, D, B, s, lda, n, me, mi, eps, reg):
    """pyipm.py:1099-1148 (the general branch; the square-Jacobian branch :1064-1097 is the same
    direction computed through inv(B) and cannot be compiled in the reference as written — its input
    list names ``s_dev`` twice, :877-880).  ``reg`` = reg_coef*eta*mu**beta (:1113).
    Returns the RAW direction (multiplier rows not yet sign-flipped, :1723-1725)."""
    q, p = n + mi, me + mi
    m = S.shape[1]
    Adiag = zeta * np.ones((n, 1))
    if mi:
        Adiag = np.concatenate([Adiag, (lda[me:] / (s + eps)).reshape(mi, 1)], axis=0)
    BT_invA = B.T @ np.diag(1.0 / Adiag.reshape(q))
    G = BT_invA @ B
    if me:
        w = np.linalg.eigh(G[:me, :me])[0]
        rcond = np.min(np.abs(w)) / np.max(np.abs(w))
        if rcond <= eps:
            G = G.copy()
            G[:me, :me] += reg * np.eye(me)
    g1, g2 = g[:q].reshape(q, 1), g[q:].reshape(p, 1)
    v00 = BT_invA @ g1
    v01 = _solve(G, v00)
    v02 = g1 / Adiag - BT_invA.T @ v01
    v03 = -_solve(G, g2)
    v04 = -BT_invA.T @ v03
    Zg = np.concatenate([v02 + v04, v01 + v03], axis=0)
    if m == 0:
        return Zg.reshape(q + p)
    W = np.concatenate([zeta * S, Y], axis=1)
    if mi:
        W = np.concatenate([W, np.zeros((mi, 2 * m))], axis=0)
    X00 = -_solve(G, (B.T @ W) / zeta)
    X01 = W / zeta + BT_invA.T @ X00
    X02 = W.T @ X01
    Minv = np.block([[zeta * SS, L], [L.T, -D]])
    v10 = W.T @ Zg[:q]
    v11 = _solve(X02 - Minv, v10)
    X10 = np.concatenate([X01, -X00], axis=0)
    return (Zg - X10 @ v11).reshape(q + p)


def direction(g, zeta, S, Y, SS, L, D, Je=None, Ji=None, s=None, lda=None, eps=np.finfo(float).eps, reg=0.0):
    """lbfgs_dir (pyipm.py:1184-1246) with the composite Jacobian of :582-607; RAW direction."""
    n = S.shape[0]
    me = 0 if Je is None else Je.shape[1]
    mi = 0 if Ji is None else Ji.shape[1]
    if me + mi == 0:
        return direction_unconstrained(g, zeta, S, Y, SS, L, D)
    cols = [c for c in (Je, Ji) if c is not None]
    top = np.concatenate(cols, axis=1)
    if mi:
        B = np.concatenate([top, np.concatenate([np.zeros((mi, me)), -np.eye(mi)], axis=1)], axis=0)
    else:
        B = top
    return direction_constrained(g, zeta, S, Y, SS, L, D, B, s, lda, n, me, mi, eps, reg)
