#!/usr/bin/env python
"""The backward sweep as one device-driven launch (k_bwd_sweep, set_option("sweep_persist", 1)) against the per-panel
launches: same direction to rounding on several shapes, repeated (a missed ordering between workgroups would show up as
an occasional difference), and the exposed solve time of both.  Diagnostics; tests/test_gpu_symmetric.py holds the test."""
import os, sys
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyipm_amd.newton import NewtonCore
from bench import make_qp_device

dev = torch.device("cuda", 0)
shapes = [(3072, 768, 1152, 256), (1000, 300, 500, 256), (900, 0, 0, 128), (2000, 500, 0, 256), (700, 0, 600, 128), (16384, 4096, 6144, 256)]
if len(sys.argv) > 1 and sys.argv[1] == "small":
    shapes = shapes[:-1]
for (n, me, mi, nb) in shapes:
    qp = make_qp_device(n, me, mi, 1, dev)
    core = NewtonCore(n, me, mi, device=0, nb=nb)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]); core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    core.set_option("profile", 1)
    res = {}
    for sp in (0, 1):
        core.set_option("sweep_persist", sp)
        dz, st = core.step(0.0, 0.0, refine=0)
        ts = []
        outs = []
        for rep in range(6 if n > 8000 else 20):
            dz, st = core.step(0.0, 0.0, refine=0)
            ts.append(core.timings()["solve_ms"]); outs.append(dz.clone())
        same = all(torch.equal(outs[0], o) for o in outs)
        res[sp] = (outs[0], sorted(ts)[len(ts) // 2], same)
    d = float((res[0][0] - res[1][0]).norm() / res[0][0].norm())
    g = core.residual(); raw = res[1][0].clone()
    if me + mi: raw[n + mi:] *= -1.0
    berr = float((core.matvec(raw) - g).norm() / g.norm())
    print("n=%d me=%d mi=%d nb=%d: rel diff %.2e  repeatable %s/%s  berr %.2e  solve %.3f -> %.3f ms" % (
        n, me, mi, nb, d, res[0][2], res[1][2], berr, res[0][1], res[1][1]), flush=True)
    # a refined solve (several sweeps per solve) through the persistent launch
    dz2, _ = core.step(0.0, 0.0, refine=2)
    print("   refine=2: finite %s, diff to unrefined %.2e" % (bool(torch.isfinite(dz2).all()), float((dz2 - res[1][0]).norm() / dz2.norm())))
    core.close()
