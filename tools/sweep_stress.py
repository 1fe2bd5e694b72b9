#!/usr/bin/env python
"""Stress of the device-driven sweeps (k_fwd_sweep / k_bwd_sweep): many solves on several shapes, every result compared bit
for bit with the first one of its kind and to rounding with the per-panel launches; a missed ordering between workgroups
would show as an occasional difference, a lost wake-up as a 2 s stall (reported through the error word / NaN)."""
import os, sys, time
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyipm_amd.newton import NewtonCore
from bench import make_qp_device
dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
shapes = [(3072, 768, 1152, 256), (1000, 300, 500, 256), (900, 0, 0, 128), (5000, 1000, 2500, 256), (700, 0, 600, 128), (16384, 4096, 6144, 256)]
for (n, me, mi, nb) in shapes:
    qp = make_qp_device(n, me, mi, 1, dev)
    core = NewtonCore(n, me, mi, device=0, nb=nb)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]); core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    core.step(0.0, 0.0)
    rhs = torch.randn(core.N, dtype=torch.float64, device=dev)
    core.set_option("sweep_persist", 0); ref = core.solve(rhs, flip=False).clone()
    core.set_option("sweep_persist", 1); first = core.solve(rhs, flip=False).clone()
    bad = 0; tmax = 0.0
    r = reps if n < 10000 else max(reps // 6, 20)
    for k in range(r):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x = core.solve(rhs, flip=False)
        torch.cuda.synchronize(); tmax = max(tmax, time.perf_counter() - t0)
        if not torch.equal(x, first): bad += 1
        if k % 7 == 3:
            core.step(0.0, 0.0)                      # a factorisation in between (its factor_end reads the error word)
    print("n=%d me=%d mi=%d nb=%d: %d solves, %d differing, slowest %.2f ms, vs per-panel %.1e" % (
        n, me, mi, nb, r, bad, tmax * 1e3, float((first - ref).norm() / ref.norm())), flush=True)
    core.close()
