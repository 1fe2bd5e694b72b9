import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_qp_device
from oracle import newton_oracle as orc
from pyipm_amd.newton import NewtonCore
from threadpoolctl import threadpool_limits
n, me, mi = [int(a) for a in sys.argv[1:4]]
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 16
N = n + 2 * mi + me
dev = torch.device("cuda", 0)
qp = make_qp_device(n, me, mi, 5, dev)
core = NewtonCore(n, me, mi, device=0)
core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
dz, st = core.step(0.0, 0.0)
dz = dz.cpu().numpy()
host = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in qp.items()}
core.close(); del qp; torch.cuda.empty_cache()
t0 = time.perf_counter()
with threadpool_limits(limits=threads):
    ref, _, H, g = orc.newton_step(host["d2L"], host["Je"], host["Ji"], host["df"], host["ce"], host["ci"], host["s"],
                                   host["lam"], host["mu"], n, me, mi, regularise=False)
print("oracle seconds", time.perf_counter() - t0, "threads", threads)
def unflip(v):
    r = v.copy(); r[n + mi:] *= -1.0; return r
for name, v in (("device", dz), ("oracle", ref)):
    r = H @ unflip(v) - g
    print(name, "berr on the host with the oracle's H: %.3e" % (np.linalg.norm(r) / np.linalg.norm(g)))
d = dz - ref
for nm, a, b in (("x", 0, n), ("s", n, n + mi), ("le", n + mi, n + mi + me), ("li", n + mi + me, N)):
    print(nm, "rel diff %.3e" % (np.linalg.norm(d[a:b]) / max(np.linalg.norm(ref[a:b]), 1e-300)))
