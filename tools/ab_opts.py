"""In-process A/B of option sets on one problem (boxes differ by +-3 %: compare inside one run only).
usage: python tools/ab_opts.py N ME MI STEPS "name=value,name=value" "..." ...      ("" = defaults)"""
import os
import sys
import time
os.environ.setdefault("PYIPM_EXPERT", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_qp_device
from pyipm_amd.newton import NewtonCore

n, me, mi, steps = (int(v) for v in sys.argv[1:5])
sets = sys.argv[5:] or [""]
dev = torch.device("cuda", 0)
qp = make_qp_device(n, me, mi, 0, dev)
res = {s: [] for s in sets}
for rep in range(3):
    for s in sets:
        core = NewtonCore(n, me, mi, device=0)
        core.set_option("profile", 1)
        for kv in [x for x in s.split(",") if x]:
            k, v = kv.split("=")
            core.set_option(k, float(v))
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        for _ in range(3):
            core.step(0.0, 0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pan = 0.0
        for _ in range(steps):
            core.step(0.0, 0.0)
            pan += core.timings()["panel_ms"]
        torch.cuda.synchronize()
        res[s].append(((time.perf_counter() - t0) / steps * 1e3, pan / steps))
        core.close()
for s in sets:
    print("%-60s %s" % (repr(s), "  ".join("%.3f (panel %.2f)" % x for x in res[s])))
