import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyipm_amd.newton import NewtonCore
from pyipm_amd.problems import make_qp
for (n, me, mi) in [(100, 0, 0), (300, 0, 0), (600, 0, 0), (700, 64, 300), (2048, 0, 2048)]:
    qp = make_qp(n, me, mi, 1)
    out = []
    for pl in (0, 1):
        core = NewtonCore(n, me, mi, device=0)
        core.set_option("bwd_diag4", pl)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        dz, st = core.step(0.0, 0.0)
        out.append(dz.cpu().numpy().copy())
        core.close()
    d = np.abs(out[0] - out[1])
    N = n + 2 * mi + me
    print((n, me, mi), "N", N, "max diff", d.max(), "rel", d.max() / np.abs(out[0]).max(), "per 256-block:", [float("%.1e" % d[i:i + 256].max()) for i in range(0, N, 256)])
