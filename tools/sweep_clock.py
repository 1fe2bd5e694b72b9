#!/usr/bin/env python
"""Phases of workgroup 0 of the one-launch backward sweep (k_bwd_sweep), per panel, from its own timestamps."""
import os, sys
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyipm_amd.newton import NewtonCore
from bench import make_qp_device
n, me, mi = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3072, 768, 1152)
dev = torch.device("cuda", 0)
qp = make_qp_device(n, me, mi, 0, dev)
core = NewtonCore(n, me, mi, device=0, nb=256)
core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]); core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
core.step(0.0, 0.0); torch.cuda.synchronize()
P = core.Npad // 256
buf = torch.zeros(4 * P + 64, dtype=torch.int64, device=dev)
core.set_option("debug_timeline_ptr", float(buf.data_ptr()))
core.solve(flip=False); torch.cuda.synchronize()
core.set_option("debug_timeline_ptr", 0.0)
d = buf.cpu().numpy()[:4 * P].reshape(P, 4).astype(np.float64) * 0.01      # us
order = np.arange(P - 1, -1, -1)
d = d[order]
wait = d[:, 1] - d[:, 0]; inpanel = d[:, 2] - d[:, 1]; pub = d[:, 3] - d[:, 2]
gap = d[1:, 0] - d[:-1, 3]
print("panels %d, sweep %.1f us from first to last stamp" % (P, d[-1, 3] - d[0, 0]))
for name, x in (("wait y/near", wait[1:]), ("in-panel", inpanel), ("store x + ack", pub), ("flag, progress poll, next top", gap)):
    print("%-32s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us" % (name, x.mean(), *np.percentile(x, [10, 50, 90])))
