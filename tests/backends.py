"""Test-only Newton backends (checkers, never shipped in pyipm_amd/)."""
import numpy as np

from oracle import newton_oracle as orc


class OracleBackend(object):
    """Implements the backend interface of pyipm_amd.ipm with the CPU oracle (reference path:
    eigen-inertia reghess + LU).  Lets the CPU suite exercise the HOST loop / line search."""

    def __init__(self, n, me, mi):
        self.n, self.me, self.mi = n, me, mi
        self.calls = []

    def direction(self, d2L, Je, Ji, df, ce, ci, s, lda, mu, delta, mu_host, eta, beta, reg_coef, delta0, eps):
        st = {}
        dz, delta, Hc, g = orc.newton_step(d2L, Je, Ji, df, ce, ci, s, lda, mu, self.n, self.me, self.mi,
                                           delta=delta, mu_host=mu_host, eta=eta, beta=beta, eps=eps, stats=st)
        self.calls.append({"Hc": Hc, "g": g, "dz": dz, "delta": delta})
        return dz, delta, st


class OracleLbfgsBackend(object):
    """L-BFGS backend interface of pyipm_amd.ipm on the CPU oracle (oracle/lbfgs_oracle.py)."""

    def __init__(self, n, me, mi):
        self.n, self.me, self.mi = n, me, mi
        self.calls = []

    def lbfgs_direction(self, Je, Ji, s, lda, g, zeta, S, Y, SS, L, D, reg, eps):
        from oracle import lbfgs_oracle as lo
        dz = lo.direction(g, zeta, S, Y, SS, L, D, Je=Je, Ji=Ji, s=s, lda=lda, eps=eps, reg=reg)
        self.calls.append({"g": np.array(g), "zeta": float(zeta), "S": np.array(S), "Y": np.array(Y),
                           "SS": np.array(SS), "L": np.array(L), "D": np.array(D), "dz_raw": np.array(dz)})
        return dz


class ModelCore(object):
    """NumPy model of ONE RANK of the HIP core's per-panel interface (block-cyclic local columns,
    pack/unpack messages, panel-wise substitutions).  Lets the world_size-2 gloo tests exercise
    pyipm_amd.dist.DistNewton on CPU.  Built on oracle/block_ldl_model.sweep_invert."""
    on_device = False
    device = "cpu"

    def __init__(self, n, me, mi, H, g, nb=128, world=1, rank=0, tb=64):
        import torch
        from oracle.block_ldl_model import sweep_invert
        self.torch, self.sweep_invert = torch, sweep_invert
        self.n, self.me, self.mi = n, me, mi
        self.N = n + 2 * mi + me
        self.Npad = ((self.N + 127) // 128) * 128
        self.nb, self.world, self.rank, self.tb = nb, world, rank, tb
        self.npanels = (self.Npad + nb - 1) // nb
        self.H, self.g = np.asarray(H, dtype=np.float64), np.asarray(g, dtype=np.float64)
        self.mine = [p for p in range(self.npanels) if p % world == rank]
        self.lcol = {}
        c = 0
        for p in self.mine:
            self.lcol[p] = c
            c += self.pw(p)
        self.ncols_local = c

    def pw(self, p):
        return min(self.nb, self.Npad - p * self.nb)

    def new_buffer(self, numel):
        return self.torch.empty(int(numel), dtype=self.torch.float64)

    def residual(self):
        return self.torch.from_numpy(self.g.copy())

    def assemble(self, delta=0.0, delta_c=0.0):
        N, Npad = self.N, self.Npad
        full = np.eye(Npad)
        full[:N, :N] = self.H
        full[:self.n, :self.n] += delta * np.eye(self.n)
        if self.me:
            i1 = self.n + self.mi
            full[i1:i1 + self.me, i1:i1 + self.me] -= delta_c * np.eye(self.me)
        self.A = np.zeros((Npad, max(self.ncols_local, 1)))
        for p in self.mine:
            c0 = p * self.nb
            blk = np.tril(full)[:, c0:c0 + self.pw(p)]
            self.A[:, self.lcol[p]:self.lcol[p] + self.pw(p)] = blk
        self.W = {}
        self.L = {}
        self.Tinv = {}
        self.Tsave = {}

    def factor_begin(self):
        self.st = dict(n_neg=0, n_zero=0, n_2x2=0, n_pos=0, nonfinite=0, d_min=1e308, d_max=0.0, growth=0.0)

    def factor_end(self):
        return dict(self.st)

    def factor_panel(self, p):
        assert p % self.world == self.rank
        tb, c0, nbw, lc = self.tb, p * self.nb, self.pw(p), self.lcol[p]
        W = np.zeros((self.Npad, nbw))
        for t in range(nbw // tb):
            j0, l0 = c0 + t * tb, lc + t * tb
            if t > 0:     # left-looking in-panel update with the tiles before this one
                self.A[j0:, l0:l0 + tb] -= self.A[j0:, lc:l0] @ W[j0:j0 + tb, :t * tb].T
            T = self.A[j0:j0 + tb, l0:l0 + tb]
            Ti, s = self.sweep_invert(T)
            self.Tinv[(p, t)] = Ti
            Tf = np.tril(T) + np.tril(T, -1).T
            self.Tsave[(p, t)] = Tf
            real = max(0, min(tb, self.N - j0))
            self.st["n_neg"] += s["neg"]; self.st["n_zero"] += s["zero"]; self.st["n_2x2"] += s["n2x2"]
            self.st["n_pos"] += real - s["neg"] if real else 0     # static pivots count by their sign (device semantics)
            self.st["d_min"] = min(self.st["d_min"], s["dmin"]); self.st["d_max"] = max(self.st["d_max"], s["dmax"])
            below = slice(j0 + tb, self.Npad)
            W[below, t * tb:(t + 1) * tb] = self.A[below, l0:l0 + tb]
            St = W[below, t * tb:(t + 1) * tb]
            Lt = St @ Ti
            Lt = Lt + (St - Lt @ Tf) @ Ti           # one refinement step of L T = S (as on the device)
            self.A[below, l0:l0 + tb] = Lt
            if Lt.size:
                self.st["growth"] = max(self.st["growth"], float(np.abs(Lt).max()))
        self.W[p] = W
        self.L[p] = None        # owner reads L from its own storage

    def panel_msg_numel(self, p):
        nbw = self.pw(p)
        m = self.Npad - (p * self.nb + nbw)
        return m * nbw + 2 * (nbw // self.tb) * self.tb * self.tb + nbw // self.tb

    def panel_pack(self, p, buf):
        nbw = self.pw(p); c1 = p * self.nb + nbw; m = self.Npad - c1
        out = buf.numpy()
        out[:m * nbw] = self.W[p][c1:, :].T.reshape(-1)             # column-major, ld = m
        nt, t2 = nbw // self.tb, self.tb ** 2
        for t in range(nt):
            out[m * nbw + t * t2: m * nbw + (t + 1) * t2] = self.Tinv[(p, t)].reshape(-1)
            out[m * nbw + (nt + t) * t2: m * nbw + (nt + t + 1) * t2] = self.Tsave[(p, t)].reshape(-1)

    def panel_unpack(self, p, buf):
        assert p % self.world != self.rank
        nbw = self.pw(p); c1 = p * self.nb + nbw; m = self.Npad - c1; tb = self.tb
        arr = buf.numpy()
        W = np.zeros((self.Npad, nbw))
        W[c1:, :] = arr[:m * nbw].reshape(nbw, m).T
        L = np.zeros((self.Npad, nbw))
        nt = nbw // tb
        for t in range(nt):
            Ti = arr[m * nbw + t * tb * tb: m * nbw + (t + 1) * tb * tb].reshape(tb, tb).copy()
            Tf = arr[m * nbw + (nt + t) * tb * tb: m * nbw + (nt + t + 1) * tb * tb].reshape(tb, tb).copy()
            self.Tinv[(p, t)], self.Tsave[(p, t)] = Ti, Tf
            St = W[c1:, t * tb:(t + 1) * tb]
            Lt = St @ Ti
            L[c1:, t * tb:(t + 1) * tb] = Lt + (St - Lt @ Tf) @ Ti
        self.W[p], self.L[p] = W, L

    def trailing_update(self, p):
        self.trailing_update_range(p, p + 1, self.npanels)

    def trailing_update_range(self, p, first, count):
        nbw = self.pw(p); c1 = p * self.nb + nbw
        own = p % self.world == self.rank
        L = self.A[:, self.lcol[p]:self.lcol[p] + nbw] if own else self.L[p]
        W = self.W[p]
        for q in self.mine:
            if q <= p or q < first or q >= first + count:
                continue
            q0, qw, lq = q * self.nb, self.pw(q), self.lcol[q]
            self.A[q0:, lq:lq + qw] -= L[q0:, :] @ W[q0:q0 + qw, :].T

    def fwd_panel(self, p, v):
        x = v.numpy(); tb = self.tb
        c0, nbw, lc = p * self.nb, self.pw(p), self.lcol[p]
        for t in range(nbw // tb):
            j0 = c0 + t * tb
            x[j0 + tb:] -= self.A[j0 + tb:, lc + t * tb: lc + (t + 1) * tb] @ x[j0:j0 + tb]

    def diag_panel(self, p, v):
        x = v.numpy(); tb = self.tb
        for t in range(self.pw(p) // tb):
            j0 = p * self.nb + t * tb
            y = x[j0:j0 + tb].copy()
            z = self.Tinv[(p, t)] @ y
            x[j0:j0 + tb] = z + self.Tinv[(p, t)] @ (y - self.Tsave[(p, t)] @ z)

    def bwd_panel(self, p, v):
        x = v.numpy(); tb = self.tb
        c0, nbw, lc = p * self.nb, self.pw(p), self.lcol[p]
        for t in range(nbw // tb - 1, -1, -1):
            j0 = c0 + t * tb
            x[j0:j0 + tb] -= self.A[j0 + tb:, lc + t * tb: lc + (t + 1) * tb].T @ x[j0 + tb:]


def lbfgs_direction_row_shard(allreduce, Je, Ji, g_loc, s, lda, zeta, S, Y, SS, L, D, me, mi):
    """NumPy model of ONE RANK of the row-sharded L-BFGS direction, with the three sums of include/pyipm_lbfgs.h
    (pyipm_lbfgs_set_allreduce) at the same places and on the same quantities as pyipm_amd/csrc/lbfgs_impl.hpp.
    Inputs are this rank's rows (Je, Ji, S, Y and the x part of g_loc = [g_x rows | g_s | g_lambda]); returns
    [dz_x rows | dz_s | dz_lambda], RAW.  `allreduce(array)` sums over the ranks in place."""
    import scipy.linalg
    n = S.shape[0]
    m = S.shape[1]
    p, r = me + mi, 2 * m
    gx, gs, gl = g_loc[:n], g_loc[n:n + mi], g_loc[n + mi:]
    if p == 0:
        W = np.concatenate([S, zeta * Y], axis=1)
        t = W.T @ gx
        allreduce(t)                                                   # (3') W'g
        if m == 0:
            return zeta * gx
        K = np.block([[np.zeros((m, m)), L], [L.T, D + zeta * SS]])
        return zeta * gx - W @ scipy.linalg.solve(K, t)
    J = np.concatenate([c for c in (Je, Ji) if c is not None], axis=1)
    sig = lda[me:] / (s + np.finfo(float).eps) if mi else np.zeros(0)
    G = J.T @ J
    allreduce(G)                                                       # (1) J'J
    G = G + np.diag(np.concatenate([np.zeros(me), zeta / sig]))        # zeta * G of the reference
    W = np.concatenate([zeta * S, Y], axis=1)
    V = np.concatenate([gx[:, None], W], axis=1)
    P = J.T @ V
    allreduce(P)                                                       # (2) J'[g_x | W]
    R = np.empty_like(P)
    R[:, 0] = P[:, 0] - zeta * gl
    if mi:
        R[me:, 0] -= zeta * gs / sig
    R[:, 1:] = -P[:, 1:]
    R = scipy.linalg.solve(G, R, assume_a="pos")
    v11 = np.zeros(0)
    if m:
        Ha = W.T @ V
        allreduce(Ha)                                                  # (3) W'[g_x | W]
        Hb = P[:, 1:].T @ R
        Hs = np.concatenate([(Ha[:, :1] - Hb[:, :1]), Ha[:, 1:] + Hb[:, 1:]], axis=1) / zeta
        Minv = np.block([[zeta * SS, L], [L.T, -D]])
        v11 = scipy.linalg.solve(Hs[:, 1:] - Minv, Hs[:, 0])
    u = R[:, 0] + (R[:, 1:] @ v11 if m else 0.0)
    dzx = (gx - (W @ v11 if m else 0.0) - J @ u) / zeta
    dzs = (gs + u[me:]) / sig if mi else np.zeros(0)
    return np.concatenate([dzx, dzs, u])
