"""NumPy model of the HIP factorisation algorithm.  TEST INFRASTRUCTURE ONLY.

A line-for-line *algorithmic* twin (not a code twin) of the device path in
``pyipm_amd/csrc``: block LDL' with ``tb x tb`` block pivots, each diagonal tile
inverted in place by symmetric sweeps with Bunch-Kaufman 1x1/2x2 pivot selection
restricted to the tile.  It lets the CPU tests (a) check that the algorithm
itself meets the <=1e-10 parity bar against the reference-following oracle
before any GPU is involved, and (b) stand in for the HIP panel operations in the
world_size-2 ``gloo`` tests of the multi-GPU host orchestration.  It is never
imported by the product.

Factorisation:  A = Lb * blockdiag(T_k) * Lb'   with  Lb unit BLOCK lower triangular,
T_k the k-th Schur-complement diagonal tile,  Lb[i,k] = S[i,k] * inv(T_k)  where
S is the Schur-complemented column block ("W" on the device).  The explicit inverse only
solves Lb T = S to cond(T)*eps, so the block solves get ``refine`` steps of
R = S - Lb T ; Lb += R inv(T)  (backward-stable rows; same in the substitutions).
Solve:  y_k = b_k - sum_{j<k} Lb[k,j] y_j ;  z_k = inv(T_k) y_k ;
        x_k = z_k - sum_{i>k} Lb[i,k]' x_i .
"""
from __future__ import annotations

import numpy as np

ALPHA = (1.0 + np.sqrt(17.0)) / 8.0       # Bunch-Kaufman threshold


SQRT_EPS = float(np.sqrt(np.finfo(np.float64).eps))


def sweep_invert(T, pivtol_rel=1e-14, anorm=0.0, neg_lim=None):
    """Invert a symmetric tile by symmetric sweeps with tile-local Bunch-Kaufman pivoting.

    Returns (Tinv, stats) with stats = dict(neg, zero, n2x2, dmin, dmax).  After
    sweeping every index the working matrix equals -inv(T); the unswept block is
    always the current Schur complement, so pivot selection is ordinary BK on it
    with no physical row/column swaps.

    Static pivots (as on the device, kernels_factor.hpp): a pivot that has cancelled below
    ``pivtol_rel`` x its column's original magnitude is replaced by ``+-sqrt(eps)*anorm``
    (``anorm`` = max |entry| of the assembled matrix, 1 when unknown), negative from tile
    index ``neg_lim`` on (the multiplier rows), counted in ``zero`` AND by its sign.
    """
    pert = SQRT_EPS * (anorm if anorm > 0.0 else 1.0)
    if neg_lim is None:
        neg_lim = np.inf
    B = np.array(T, dtype=np.float64)
    B = np.tril(B) + np.tril(B, -1).T
    tb = B.shape[0]
    unswept = np.ones(tb, dtype=bool)
    colmax0 = np.max(np.abs(B), axis=0) if tb else np.zeros(0)   # reference scale of each pivot (as on the device)
    neg = zero = n2 = 0
    dmin, dmax = np.inf, 0.0

    def sweep1(p):
        nonlocal neg, zero, dmin, dmax
        d = B[p, p]
        ad = abs(d)
        pivtol = pivtol_rel * colmax0[p]
        if ad <= pivtol:
            zero += 1
            t = max(pivtol, pert)
            d = t if p < neg_lim else -t
            if d < 0:
                neg += 1
        else:
            if d < 0:
                neg += 1
            dmin, dmax = min(dmin, ad), max(dmax, ad)
        col = B[:, p].copy()
        colp = col / d
        B[:, :] -= np.outer(colp, col)
        B[:, p] = colp
        B[p, :] = colp
        B[p, p] = -1.0 / d
        unswept[p] = False

    def sweep2(p, q):
        nonlocal neg, zero, n2, dmin, dmax
        a, b, c = B[p, p], B[p, q], B[q, q]
        det = a * c - b * b
        # BK guarantees det < 0: one positive, one negative eigenvalue
        n2 += 1
        tr = a + c
        disc = np.sqrt((a - c) ** 2 + 4 * b * b)
        e1, e2 = 0.5 * (tr + disc), 0.5 * (tr - disc)
        pivtol = pivtol_rel * max(colmax0[p], colmax0[q])
        neg += 1
        for e in (e1, e2):
            if abs(e) <= pivtol:
                zero += 1
            else:
                dmin, dmax = min(dmin, abs(e)), max(dmax, abs(e))
        if not abs(det) > 0.0:
            det = -pert * pert
        ia, ib, ic = c / det, -b / det, a / det           # inverse of [[a,b],[b,c]]
        cp, cq = B[:, p].copy(), B[:, q].copy()
        lp = cp * ia + cq * ib
        lq = cp * ib + cq * ic
        B[:, :] -= np.outer(lp, cp) + np.outer(lq, cq)
        B[:, p] = lp
        B[p, :] = lp
        B[:, q] = lq
        B[q, :] = lq
        B[p, p], B[p, q], B[q, p], B[q, q] = -ia, -ib, -ib, -ic
        unswept[p] = unswept[q] = False

    while unswept.any():
        idx = np.flatnonzero(unswept)
        p = idx[0]                       # standard BK candidate: first unswept index (as on the device)
        app = abs(B[p, p])
        if idx.size == 1:
            sweep1(p)
            continue
        others = idx[idx != p]
        colmag = np.abs(B[others, p])
        r = others[int(np.argmax(colmag))]
        lam = abs(B[r, p])
        if lam == 0.0 or app >= ALPHA * lam:
            sweep1(p)
            continue
        rest = idx[idx != r]
        sigma = np.max(np.abs(B[r, rest]))
        if app * sigma >= ALPHA * lam * lam:
            sweep1(p)
        elif abs(B[r, r]) >= ALPHA * sigma:
            sweep1(r)
        else:
            sweep2(p, r)
    return -B, dict(neg=neg, zero=zero, n2x2=n2, dmin=dmin, dmax=dmax)


class BlockLDL(object):
    """Dense block-LDL' of a symmetric matrix with tile size ``tb`` (device: 64)."""

    def __init__(self, A, tb=64, nreal=None, refine=1, neg_from=None, sigma_from=None):
        """``neg_from``: index from which pivots are expected negative (n + mi of a KKT matrix): the sign a static
        pivot takes.  ``stats['zero']`` counts static pivots; they are also in neg / (N - neg) by their sign.
        ``sigma_from``: first index of the slack block (n): its diagonal, Sigma, stays out of the scale of a static pivot
        as on the device (k_assemble)."""
        A = np.array(A, dtype=np.float64)
        N = A.shape[0]
        self.A0 = np.tril(A) + np.tril(A, -1).T
        B = np.abs(A)
        if sigma_from is not None and neg_from is not None and A.size:
            idx = np.arange(int(sigma_from), int(min(neg_from, N)))
            B = B.copy(); B[idx, idx] = 0.0
        anorm = float(B.max()) if A.size else 0.0
        if neg_from is None:
            neg_from = np.inf
        self.N = N
        self.tb = tb
        Np = ((N + tb - 1) // tb) * tb
        self.Np = Np
        M = np.eye(Np)
        M[:N, :N] = np.tril(A) + np.tril(A, -1).T
        self.nt = Np // tb
        self.Tinv = []
        self.T = []
        self.refine_tile = []
        self.refine = refine            # refinement steps of the block solves L T = S and T z = y (device: block_refine)
        self.stats = dict(neg=0, zero=0, n2x2=0, dmin=np.inf, dmax=0.0)
        tb_ = tb
        for k in range(self.nt):
            k0, k1 = k * tb_, (k + 1) * tb_
            T = np.tril(M[k0:k1, k0:k1]) + np.tril(M[k0:k1, k0:k1], -1).T
            Ti, st = sweep_invert(T, anorm=anorm, neg_lim=neg_from - k0)
            self.Tinv.append(Ti)
            self.T.append(T)
            self.refine_tile.append(st["zero"] == 0)      # a statically pivoted tile has nothing to refine against (device: Tflag = 0)
            for key in ("neg", "zero", "n2x2"):
                self.stats[key] += st[key]
            self.stats["dmin"] = min(self.stats["dmin"], st["dmin"])
            self.stats["dmax"] = max(self.stats["dmax"], st["dmax"])
            if k1 < Np:
                W = M[k1:, k0:k1].copy()
                L = W @ Ti
                for _ in range(refine if st["zero"] == 0 else 0):
                    L = L + (W - L @ T) @ Ti
                M[k1:, k1:] -= L @ W.T
                M[k1:, k0:k1] = L
        # padding rows contribute positive unit pivots only
        self.M = M

    def solve(self, b):
        tb, nt, Np = self.tb, self.nt, self.Np
        y = np.zeros(Np)
        y[:self.N] = b
        for k in range(nt):
            k0, k1 = k * tb, (k + 1) * tb
            if k1 < Np:
                y[k1:] -= self.M[k1:, k0:k1] @ y[k0:k1]
        for k in range(nt):
            k0, k1 = k * tb, (k + 1) * tb
            yk = y[k0:k1].copy()
            z = self.Tinv[k] @ yk
            for _ in range(self.refine if self.refine_tile[k] else 0):
                z = z + self.Tinv[k] @ (yk - self.T[k] @ z)
            y[k0:k1] = z
        for k in range(nt - 1, -1, -1):
            k0, k1 = k * tb, (k + 1) * tb
            if k1 < Np:
                y[k0:k1] -= self.M[k1:, k0:k1].T @ y[k1:]
        return y[:self.N]

    def solve_refined(self, b, target=1e-14, maxit=8):
        """The adaptive refinement of pyipm_newton_solve(refine < 0): r = b - A x from the UNfactored matrix,
        x += solve(r), until |r|/|b| <= target, a step gains less than 4x, or maxit.  Returns (x, info)."""
        b = np.asarray(b, dtype=np.float64)
        x = self.solve(b)
        bn = np.linalg.norm(b)
        prev, info = -1.0, dict(steps=0, converged=False, backward_error0=-1.0, backward_error=-1.0)
        for it in range(maxit + 1):
            r = b - self.A0 @ x
            berr = float(np.linalg.norm(r) / bn) if bn > 0 else float(np.linalg.norm(r))
            if it == 0:
                info["backward_error0"] = berr
            info["backward_error"] = berr
            if not np.isfinite(berr):
                break
            if berr <= target:
                info["converged"] = True
                break
            if it == maxit or (prev >= 0 and berr > 0.25 * prev):
                break
            prev = berr
            x = x + self.solve(r)
            info["steps"] = it + 1
        return x, info
