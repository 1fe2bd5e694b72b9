#!/usr/bin/env python
"""Latency of the panel chain: factor a small system (dominated by k_tile_invert) and report us per tile."""
import os, sys, time
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyipm_amd.newton import NewtonCore
from pyipm_amd.problems import make_qp
n, me, mi = 1024, 256, 384      # N = 2048 -> 32 tiles
qp = make_qp(n, me, mi, 1)
core = NewtonCore(n, me, mi, device=0)
core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]); core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
core.set_option("lookahead", 0)
if len(sys.argv) > 1:
    core.set_option("tile_blocked", float(sys.argv[1]))
ts = []
for rep in range(6):
    core.assemble(0.0, 0.0); torch.cuda.synchronize()
    t0 = time.perf_counter(); core.factor(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
print("N=%d factor median %.1f us -> %.1f us per 64-tile (incl. scale/in-panel/updates)" % (core.N, ts[3] * 1e6, ts[3] * 1e6 / (core.Npad / 64)))
