"""Device-side timeline of one factorisation by groups (PYIPM_GROUP_TRACE: HIP events at the begin / end of every group's chain + rows,
head and bulk update; no tracer on the host).  usage: python tools/group_trace.py N ME MI ["opt=value,..."]"""
import os, sys
os.environ["PYIPM_GROUP_TRACE"] = "1"
os.environ.setdefault("PYIPM_EXPERT", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_qp_device
from pyipm_amd.newton import NewtonCore
n, me, mi = (int(v) for v in sys.argv[1:4])
qp = make_qp_device(n, me, mi, 0, torch.device("cuda", 0))
core = NewtonCore(n, me, mi, device=0)
for kv in [x for x in (sys.argv[4] if len(sys.argv) > 4 else "").split(",") if x]:
    k, v = kv.split("="); core.set_option(k, float(v))
core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]); core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
for rep in range(4):
    sys.stderr.write("---- step %d\n" % rep); sys.stderr.flush()
    core.step(0.0, 0.0)
torch.cuda.synchronize()
