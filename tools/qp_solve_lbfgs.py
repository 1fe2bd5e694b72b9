"""Cost of the device-resident L-BFGS loop (pyipm_amd/qp.py, lbfgs=m) at a size the exact-Hessian step cannot
hold:  min 1/2 x'(diag(d) + F F')x + c'x  s.t.  Ax = b, Gx - h >= 0,  n in the 1e5 range.
By default a FIXED budget of niter*miter = 20 iterations: the reference's L-BFGS interior-point iteration is slow to
converge on such problems (hundreds of iterations already at n = 60 for Ktol = 1e-6, no convergence within the
default 600 at n = 4096), which is a property of the algorithm reproduced here, not of the kernels; what this tool
measures is seconds per iteration and where they go.
Usage: python tools/qp_solve_lbfgs.py [--nvar N --neq ME --nineq MI --lbfgs M --niter A --miter B --ktol T]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nvar", type=int, default=262144)
    ap.add_argument("--neq", type=int, default=1024)
    ap.add_argument("--nineq", type=int, default=3072)
    ap.add_argument("--lbfgs", type=int, default=8)
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--ktol", type=float, default=1e-4)
    ap.add_argument("--niter", type=int, default=2)
    ap.add_argument("--miter", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    import torch
    from pyipm_amd.qp import QPDeviceIPM
    dev = torch.device("cuda", 0)
    n, me, mi = a.nvar, a.neq, a.nineq
    gen = torch.Generator(device=dev); gen.manual_seed(a.seed)
    rn = lambda *shape: torch.randn(shape, generator=gen, dtype=torch.float64, device=dev)      # noqa: E731
    ru = lambda k, lo, hi: lo + (hi - lo) * torch.rand(k, generator=gen, dtype=torch.float64, device=dev)   # noqa: E731
    d = ru(n, 0.5, 2.0)
    F = rn(n, a.rank) / np.sqrt(a.rank)
    c = rn(n)
    Je = rn(n, me) / np.sqrt(n) if me else None          # = A'
    Ji = rn(n, mi) / np.sqrt(n) if mi else None          # = G'
    b = 0.1 * rn(me) if me else None                     # x0 = 0 violates the equalities by N(0, 0.1)
    h = -ru(mi, 0.5, 2.0) if mi else None                # ci(0) = -h in (0.5, 2): strictly feasible slacks
    lda0 = torch.cat([torch.zeros(me, dtype=torch.float64, device=dev), torch.ones(mi, dtype=torch.float64, device=dev)])
    ipm = QPDeviceIPM(("diag+lowrank", d, F), c, Je=Je, b=b, Ji=Ji, h=h, lda0=lda0 if (me or mi) else None,
                      Ktol=a.ktol, niter=a.niter, miter=a.miter, verbosity=0, lbfgs=a.lbfgs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, s, lam, f, kkt = ipm.solve()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm = ipm.lb.last_timings()
    # first-order optimality straight from the problem data
    r = d * x + F @ (F.t() @ x) + c
    if me:
        r = r - Je @ lam[:me]
    if mi:
        r = r - Ji @ lam[me:]
    out = {"workload": "device-resident L-BFGS QP solve n=%d me=%d mi=%d, Q = diag + rank %d, lbfgs=%d, seed %d, Ktol %g" % (
               n, me, mi, a.rank, a.lbfgs, a.seed, a.ktol),
           "signal": ipm.signal, "iterations": ipm.iter_count, "solve_seconds": dt,
           "direction_seconds": ipm.timings["newton_s"], "search_seconds": ipm.timings["search_s"],
           "merit_evaluations": ipm.timings["n_phi"], "gram_launches": tm["gram_launches"],
           "last_direction_ms": tm["total_ms"], "kkt_norms": list(kkt), "fval": f,
           "stationarity_from_data": float(r.norm()),
           "eq_violation": float((x @ Je - b).norm()) if me else 0.0,
           "min_slack": float((x @ Ji - h).min()) if mi else None,
           "jacobian_GB": n * (me + mi) * 8 / 1e9}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
