#!/usr/bin/env python
"""Time one L-BFGS search direction (include/pyipm_lbfgs.h) on one MI355X and check it.

    python tools/bench_lbfgs.py --n 262144 --me 1024 --mi 3072 --m 8

Inputs are synthetic and generated on the device (J ~ N(0,1)/sqrt(n), storage from random displacement pairs
of positive curvature).  The check is size-independent: the direction must satisfy H dz = g for
H = Z - U inv(M) U' (the matrix of pyipm.py:1036-1052), applied matrix-free with torch as the checker.
The CPU comparison (the reference's arithmetic, oracle/lbfgs_oracle.py) lives in bench.py --extras: tools do not touch oracle/.
Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def storage(n, m, seed, constrained=True):
    """(zeta, S, Y, SS, L, D) with the invariants of lbfgs_update (pyipm.py:1282-1371): SS = S'S, L = strictly
    lower S'Y for constrained problems; SS = Y'Y, L = upper-triangular S'Y for unconstrained ones."""
    rng = np.random.default_rng(seed)
    S = rng.standard_normal((n, m)) / np.sqrt(n)
    Mq = rng.standard_normal((n, 16)) / 4.0
    Y = Mq @ (Mq.T @ S) + 0.5 * S                       # dg = Q dx, Q SPD
    SY = S.T @ Y
    D = np.diag(np.diag(SY))
    if constrained:
        SS, L = S.T @ S, np.tril(SY, -1)
        zeta = float(SY[-1, -1] / SS[-1, -1]) if m else 1.0
    else:
        SS, L = Y.T @ Y, np.triu(SY)
        zeta = float(SY[-1, -1] / SS[-1, -1]) if m else 1.0
    return zeta, S, Y, SS, L, D


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=262144)
    ap.add_argument("--me", type=int, default=1024)
    ap.add_argument("--mi", type=int, default=3072)
    ap.add_argument("--m", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-n", type=int, default=0, help="ignored (kept for old command lines): the CPU leg is bench.py --extras")
    a = ap.parse_args()
    import torch
    from pyipm_amd.lbfgs import LbfgsCore
    n, me, mi, m = a.n, a.me, a.mi, a.m
    p, N = me + mi, n + 2 * mi + me
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    J = torch.randn((n, p), generator=gen, dtype=torch.float64, device=dev) / np.sqrt(n) if p else None
    zeta, S, Y, SS, L, D = storage(n, m, 1, bool(p))
    rng = np.random.default_rng(2)
    s = rng.uniform(0.5, 2.0, mi)
    lda = np.concatenate([rng.standard_normal(me), rng.uniform(0.5, 2.0, mi)])
    g = rng.standard_normal(N)
    core = LbfgsCore(n, me, mi, max(m, 1), device=0)
    t0 = time.perf_counter()
    if p:
        core.stage_jacobian(J[:, :me] if me else None, J[:, me:] if mi else None)
    torch.cuda.synchronize()
    t_stage = time.perf_counter() - t0
    td = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)          # noqa: E731
    gd, sd, ld, Sd, Yd = td(g), td(s), td(lda), td(S), td(Y)
    def timed(restage):
        rec = []
        for it in range(a.reps + 1):
            if restage and p:                        # a changed Jacobian: J'J is recomputed by the next direction
                core.stage_jacobian(J[:, :me] if me else None, J[:, me:] if mi else None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = core.direction(gd, sd, ld, zeta, Sd, Yd, SS, L, D, reg=1e-12)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            if it:
                tm = core.last_timings(); tm["wall_ms"] = wall
                rec.append(tm)
        return out, {k: float(np.median([r[k] for r in rec])) for k in rec[0]}

    (dz, st), med = timed(True)
    reuse = timed(False)[1] if p else None           # linear constraints: J'J of the staged Jacobians is reused
    # ---- checker: H dz = g, matrix-free (constrained); dz = Hinv g recomputed with torch (unconstrained)
    x, ds, dl = dz[:n], dz[n:n + mi], dz[n + mi:]
    res = torch.empty_like(dz)
    if not p:
        W = torch.cat([Sd, zeta * Yd], dim=1)
        Ld, DS = td(L), td(D + zeta * SS)
        t = W.T @ gd
        Bc = -torch.linalg.solve(Ld, t[:m])
        Ac = -torch.linalg.solve(Ld.T, DS @ Bc) - torch.linalg.solve(Ld.T, t[m:])
        ref = zeta * gd + W @ torch.cat([Ac, Bc]) if m else zeta * gd
        top = gd + (ref - dz)                       # so that res - g = ref - dz
    else:
        W = torch.cat([zeta * Sd, Yd], dim=1)
        Minv = td(np.block([[zeta * SS, L], [L.T, -D]]))
        top = zeta * x - W @ torch.linalg.solve(Minv, W.T @ x) if m else zeta * x
    if p:
        top = top + J @ dl
        sig = ld[me:] / (sd + np.finfo(float).eps)
        res[n:n + mi] = sig * ds - dl[me:]
        low = J.T @ x
        low[me:] -= ds
        res[n + mi:] = low
    res[:n] = top
    check = float((res - gd).norm() / gd.norm())
    rr = 2 * m + 1
    chunks = (rr + 16) // 17
    out = {"workload": "L-BFGS direction (pyipm.py:1184-1246), QP-shaped synthetic", "n": n, "me": me, "mi": mi,
           "m": m, "dtype": "f64", "ms": med, "ms_reusing_gram": reuse, "stage_jacobian_ms": t_stage * 1e3, "stats": st,
           "residual_H_dz_minus_g_rel": check}
    if p:
        out["gram_tflops"] = med["gram_flops"] / (med["gram_ms"] * 1e-3) / 1e12 if med["gram_ms"] > 0 else None
        out["jacobian_pass_GBps"] = (chunks + 1) * n * ((p + 127) // 128 * 128) * 8 / (med["jacobian_passes_ms"] * 1e-3) / 1e9
        out["jacobian_bytes"] = n * p * 8
    print(json.dumps(out))


if __name__ == "__main__":
    main()
