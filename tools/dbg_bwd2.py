import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyipm_amd.newton import NewtonCore
from pyipm_amd.problems import make_qp
def run(n, me, mi, opts, fused):
    qp = make_qp(n, me, mi, 8)
    out = []
    for pl in (0, 1):
        core = NewtonCore(n, me, mi, device=0)
        for k, v in opts.items(): core.set_option(k, v)
        core.set_option("bwd_diag4", pl)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        if fused:
            dz, st = core.step(0.0, 0.0)
        else:
            core.residual(); core.assemble(0.0, 0.0); core.factor(); dz = core.solve(flip=True)
        out.append(dz.cpu().numpy().copy()); core.close()
    d = np.abs(out[0] - out[1]); N = len(d)
    print((n, me, mi), opts, "fused" if fused else "explicit", "max diff %.2e" % d.max(), [float("%.0e" % d[i:i + 256].max()) for i in range(0, N, 256)])
run(700, 200, 300, {}, True)
run(700, 200, 300, {}, False)
run(700, 200, 300, {"condensed": 1}, True)
run(700, 200, 300, {"condensed": 1}, False)
run(700, 0, 0, {}, False)
