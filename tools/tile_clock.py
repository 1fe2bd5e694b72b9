import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyipm_amd.newton import NewtonCore
from pyipm_amd.problems import make_qp
n, me, mi = 1024, 256, 384
qp = make_qp(n, me, mi, 1)
core = NewtonCore(n, me, mi, device=0)
core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]); core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
core.set_option("lookahead", 0)
buf = torch.zeros(16, dtype=torch.int64, device="cuda")
for rep in range(3):
    core.assemble(0.0, 0.0); core.factor()
core.set_option("debug_timeline_ptr", float(buf.data_ptr()))
core.assemble(0.0, 0.0); core.factor(); torch.cuda.synchronize()
core.set_option("debug_timeline_ptr", 0.0)
c, w, k = buf[:3].tolist()
print("last tile kernel: %d shader cycles, %d wall ticks (%.2f us) -> %.0f MHz, %.0f cycles per sweep" % (c, w, w * 0.01, c / (w * 0.01), c / 64.0))
