#!/usr/bin/env python
"""Time ONE trailing update (k_update<128>) in isolation: panel 0 factored, then trailing_update(0) repeated.
usage: python tools/bench_update.py [n me mi] [nb]   (PYIPM_NEWTON_LIB selects an ablation build)"""
import os, sys
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyipm_amd.newton import NewtonCore
from bench import make_qp_device

n, me, mi = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16384, 4096, 6144)
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda", 0)
qp = make_qp_device(n, me, mi, 0, dev)
core = NewtonCore(n, me, mi, device=0, nb=nb)
core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
for kv in sys.argv[5:]:
    k, v = kv.split("="); core.set_option(k, float(v))
core.assemble(0.0, 0.0)
core.factor_begin()
core.factor_panel(0)
m = core.Npad - nb
flops = 2.0 * nb * (m * (m + 1) / 2)
for rep in range(2):
    core.trailing_update(0)          # (repeating the update only perturbs values; timing is what matters)
torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); core.trailing_update(0); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
print("N=%d nb=%d m=%d  median %.3f ms  -> %.2f TF/s  (min %.3f ms %.2f TF/s)" %
      (core.N, nb, m, ts[2], flops / ts[2] / 1e9, ts[0], flops / ts[0] / 1e9))
