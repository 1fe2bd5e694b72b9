#!/usr/bin/env python
"""Throughput of BASELINE.json configs[4]: 512 independent QPs (n=256, mi=256 -> N=768 each), full and condensed form."""
import os, sys, time
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyipm_amd.batched import BatchedNewton

B, n, me, mi = (int(sys.argv[1]) if len(sys.argv) > 1 else 512), 256, 0, 256
gen = torch.Generator(device="cuda").manual_seed(0)
M = torch.randn(B, n, n, dtype=torch.float64, device="cuda", generator=gen)
Q = M @ M.transpose(1, 2) / n + torch.eye(n, dtype=torch.float64, device="cuda")
G = torch.randn(B, mi, n, dtype=torch.float64, device="cuda", generator=gen) / n ** 0.5
Ji = G.transpose(1, 2).contiguous()
df = torch.randn(B, n, dtype=torch.float64, device="cuda", generator=gen)
s = torch.rand(B, mi, dtype=torch.float64, device="cuda", generator=gen) * 1.5 + 0.5
lam = torch.rand(B, mi, dtype=torch.float64, device="cuda", generator=gen) * 1.5 + 0.5
ci = s + 0.1 * torch.randn(B, mi, dtype=torch.float64, device="cuda", generator=gen)
for name, kw, blocked in (("full, tile_blocked=1", {}, 1), ("full, tile_blocked=0", {}, 0), ("condensed, tile_blocked=0", {"condensed": True, "guard": False}, 0),
                          ("condensed, tile_blocked=1", {"condensed": True, "guard": False}, 1), ("condensed + guard", {"condensed": True}, 0)):
    bn = BatchedNewton(n, me, mi, **kw)
    bn.set_option("tile_blocked", blocked)
    for _ in range(2):
        bn.step_all(Q, None, Ji, df, None, ci, s, lam)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        dz, st = bn.step_all(Q, None, Ji, df, None, ci, s, lam)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    ok = all(x["n_neg"] == mi and x["n_zero"] == 0 for x in st)
    print("%-28s %d problems in %.2f ms -> %.0f Newton steps/s (inertia ok: %s); kernels %s; max backward error %.1e"
          % (name + ":", B, dt * 1e3, B / dt, ok, {k: round(v, 3) for k, v in bn.last_ms().items()}, float(bn.backward_errors(dz).max())))
    bn.close()
