#!/usr/bin/env python
"""k_fwd_sweep + k_bwd_sweep (sweep_persist=1) against the per-panel launches on solves with a right-hand side of their own
(forward pass not fused under a factorisation): difference, repeatability, time per solve."""
import os, sys, time
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyipm_amd.newton import NewtonCore
from bench import make_qp_device
dev = torch.device("cuda", 0)
shapes = [(3072, 768, 1152, 256), (1000, 300, 500, 256), (900, 0, 0, 128), (2000, 500, 0, 256), (700, 0, 600, 128), (2048, 0, 2048, 256)]
if len(sys.argv) > 1 and sys.argv[1] == "big":
    shapes.append((16384, 4096, 6144, 256))
for (n, me, mi, nb) in shapes:
    qp = make_qp_device(n, me, mi, 1, dev)
    core = NewtonCore(n, me, mi, device=0, nb=nb)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]); core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    core.step(0.0, 0.0)
    rhs = torch.randn(core.N, dtype=torch.float64, device=dev)
    res = {}
    for sp in (0, 1):
        core.set_option("sweep_persist", sp)
        outs = []; ts = []
        for rep in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            x = core.solve(rhs, flip=False)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            outs.append(x.clone())
        res[sp] = (outs[0], sorted(ts)[len(ts) // 2], all(torch.equal(outs[0], o) for o in outs))
    d = float((res[0][0] - res[1][0]).norm() / res[0][0].norm())
    r = float((core.matvec(res[1][0]) - rhs).norm() / rhs.norm())
    print("n=%d me=%d mi=%d nb=%d: rel diff %.2e  repeatable %s/%s  residual %.2e  solve(rhs) %.3f -> %.3f ms" % (
        n, me, mi, nb, d, res[0][2], res[1][2], r, res[0][1], res[1][1]), flush=True)
    core.close()
