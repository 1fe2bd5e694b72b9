#!/usr/bin/env python
"""Blocked tile inversion (tile_blocked.hpp) against the single-sweep path and NumPy, tile by tile: a KKT system with
n <= 64, me = mi = 0 IS one diagonal tile, so step() = tile inversion + one product.  Prints, per tile family, the error
of both paths against numpy.linalg.solve, the inertia against eigvalsh, and the shader cycles of the tile kernel."""
import os, sys
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyipm_amd.newton import NewtonCore

rng = np.random.default_rng(0)


def spd(n, cond=None):
    M = rng.standard_normal((n, n))
    if cond is None:
        return M @ M.T / n + np.eye(n)
    Q, _ = np.linalg.qr(M)
    return (Q * np.logspace(0, np.log10(cond), n)) @ Q.T


def quasi(n, k):
    H = spd(k); D = np.diag(rng.uniform(0.5, 2.0, n - k)); J = rng.standard_normal((k, n - k)) / np.sqrt(k)
    return np.block([[H, J], [J.T, -D]])


def tiles():
    yield "spd", spd(64)
    yield "spd n=48 (padded)", spd(48)
    yield "spd n=17", spd(17)
    yield "negdef", -spd(64)
    yield "quasi 40+24", quasi(64, 40)
    yield "quasi 16+48", quasi(64, 16)
    yield "spd cond 1e8", spd(64, 1e8)
    yield "spd cond 1e11", spd(64, 1e11)
    S = np.diag(np.logspace(-8, 8, 64)); E = rng.standard_normal((64, 64)) * 1e-9
    yield "diag 1e-8..1e8 + tiny dense", S + (E + E.T)
    Z = rng.standard_normal((64, 64)); Z = Z + Z.T; np.fill_diagonal(Z, 0.0)
    yield "zero diagonal (2x2 pivots)", Z
    A = spd(64); A[40, 40] = 0.0; A[40, :40] *= 1e-3; A[:40, 40] *= 1e-3
    yield "spd with a zero diagonal entry at 40 (fast path leaves in block 2)", A
    B = quasi(64, 32); B[50, 50] = 1e-30
    yield "quasi with a tiny pivot at 50", B
    G = rng.standard_normal((64, 64)); G = G + G.T
    yield "random symmetric indefinite", G


def run(name, H):
    n = H.shape[0]
    g = rng.standard_normal(n)
    ref = np.linalg.solve(H, g)
    w = np.linalg.eigvalsh(H)
    out = [name.ljust(62), "cond %.1e neg %2d |" % (abs(w).max() / abs(w).min(), int((w < 0).sum()))]
    for blocked in (1, 0):
        core = NewtonCore(n, 0, 0, device=0)
        core.set_option("tile_blocked", blocked)
        buf = torch.zeros(16, dtype=torch.int64, device="cuda")
        core.stage_blocks(np.triu(H) + np.triu(H, 1).T, None, None)
        core.stage_vectors(-g, None, None, None, np.zeros(0), mu=0.1)      # residual = -df = g
        core.set_option("debug_timeline_ptr", float(buf.data_ptr()))
        dz, st = core.step(0.0, 0.0)
        torch.cuda.synchronize()
        cyc = int(buf[0])
        dz = dz.cpu().numpy()
        err = np.linalg.norm(dz - ref) / np.linalg.norm(ref)
        berr = np.linalg.norm(H @ dz - g) / np.linalg.norm(g)
        out.append("%s err %.1e berr %.1e neg %2d zero %d 2x2 %2d cyc %6d |" % ("blk" if blocked else "old", err, berr, st["n_neg"],
                                                                          st["n_zero"], st["n_2x2"], cyc))
    print(" ".join(out), flush=True)


if __name__ == "__main__":
    for name, H in tiles():
        run(name, H)
