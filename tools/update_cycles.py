#!/usr/bin/env python
"""In-situ main-loop cycles of the bulk update tiles: one regular step with the update kernel's per-block stamps on; histogram of
shader cycles per tile main loop against what its MFMAs need (a 128 x 256 tile at K: 8 waves x 4 K MFMAs of 64 cycles on 4 SIMDs
= 512 K cycles per SIMD), and of the wall-clock gaps.  usage: python tools/update_cycles.py [opt=value ...]"""
import os, sys
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyipm_amd.newton import NewtonCore
from bench import make_qp_device
n, me, mi = 16384, 4096, 6144
dev = torch.device("cuda", 0)
qp = make_qp_device(n, me, mi, 0, dev)
core = NewtonCore(n, me, mi, device=0, nb=256)
core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]); core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
for kv in sys.argv[1:]:
    k, v = kv.split("="); core.set_option(k, float(v))
for _ in range(3):
    core.step(0.0, 0.0)
nrec = (core.Npad // 128) ** 2 + 4096
tl = torch.zeros(nrec * 8, dtype=torch.int64, device=dev)
core.set_option("debug_timeline_ptr", float(tl.data_ptr()))
core.step(0.0, 0.0); torch.cuda.synchronize()
core.set_option("debug_timeline_ptr", 0.0)
rec = tl.cpu().numpy().reshape(-1, 8)
rec = rec[rec[:, 0] != 0]
loop_us = (rec[:, 2] - rec[:, 1]) * 0.01
pro_us = (rec[:, 1] - rec[:, 0]) * 0.01
cyc = (rec[:, 4] >> 16).astype(np.float64)
print("records (last writer per tile-list index): %d" % len(rec))
# clusters by main-loop cycles: K = 2048 wide ~ 1.05e6+, K = 1024 (wide or narrow) ~ 0.52e6+, shorter = heads / small K
for lo, hi, name, ideal in ((0.9e6, 3e6, "K=2048 (128 x 256 tiles)", 1048576.0), (0.45e6, 0.9e6, "K=1024", 524288.0), (0.2e6, 0.45e6, "K=512", 262144.0)):
    m = (cyc >= lo) & (cyc < hi)
    if m.sum() == 0:
        continue
    r = cyc[m] / ideal
    print("%-26s %6d tiles: main loop = %.3f x the cycles its MFMAs need (p10 %.3f, p50 %.3f, p90 %.3f); %.0f MHz; prologue %.1f us (p90 %.1f)"
          % (name, m.sum(), r.mean(), *np.percentile(r, [10, 50, 90]), np.median(cyc[m] / loop_us[m]), pro_us[m].mean(), np.percentile(pro_us[m], 90)))
