#!/usr/bin/env python
"""Per-panel schedule with wide panels (nb = 512 / 1024) factored as blocks of 256-column sub-panels (factor_wide_panel):
the direction must be bit-identical to the all-stages-in-one-launch form (wide_sub=0) and to the single-rank group schedule
at nb = 256.  Diagnostics; the test proper is tests/test_gpu_dist.py::test_wide_panels_match_bitwise."""
import os, sys
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from pyipm_amd.newton import NewtonCore
from pyipm_amd.dist import DistNewton
from bench import make_qp_device

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", device_id=dev, rank=0, world_size=1)
for (n, me, mi) in ((4096, 1024, 1536), (3000, 700, 1100), (2048, 0, 0)):
    qp = make_qp_device(n, me, mi, 3, dev)
    def run(nb, wide_sub, use_dist):
        core = NewtonCore(n, me, mi, device=0, nb=nb)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]); core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        core.set_option("wide_sub", wide_sub)
        core.set_option("sweep_persist", 0)          # (the per-panel sweeps on both sides: the one-launch backward sweep sums in another order)
        if use_dist:
            drv = DistNewton(core, native=True)
            dz, st = drv.step(0.0, 0.0, refine=0)
        else:
            dz, st = core.step(0.0, 0.0, refine=0)
        dz = dz.clone(); core.close()
        return dz, st
    ref, st0 = run(256, 256, False)
    for nb in (512, 1024):
        a, sa = run(nb, 0, True)
        b, sb = run(nb, 256, True)
        c, sc = run(nb, 128, True)
        print("n=%d me=%d mi=%d nb=%d: wide vs one-launch %s (max diff %.3e), wide vs fused nb=256 %s (%.3e), sub128 vs fused %s, inertia %s %s" % (
            n, me, mi, nb, torch.equal(a, b), float((a - b).abs().max()), torch.equal(b, ref), float((b - ref).abs().max()),
            torch.equal(c, ref), sa.get("n_neg"), sb.get("n_neg")))
dist.destroy_process_group()
