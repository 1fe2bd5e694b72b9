"""The provider's products in isolation (pyipm_newton_block_products / _t, kkt_matvec): HIP-event time, HBM rate, and a check
against torch on the symmetrised triangle.  usage: python tools/bench_provider.py [n me mi]"""
import json
import os
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyipm_amd.newton import NewtonCore

n, me, mi = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 4096, 6144)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
Q = torch.randn(n, n, dtype=torch.float64, device=dev, generator=g)          # only the upper triangle is read
Je = torch.randn(n, me, dtype=torch.float64, device=dev, generator=g)
Ji = torch.randn(n, mi, dtype=torch.float64, device=dev, generator=g)
v = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
core = NewtonCore(n, me, mi, device=0, provider_only=True)
core.stage_blocks(Q, Je, Ji)
core.set_option("profile", 1)
best = None
for _ in range(7):
    q, e, i = core.block_products(v)
    st = core.provider_stats()
    if best is None or st["products_ms"] < best["products_ms"]:
        best = st
S = torch.triu(Q) + torch.triu(Q, 1).t()
ref = S @ v
err = float((q - ref).abs().max() / ref.abs().max())
qo = core.block_products(v, want=(True, False, False))[0]
only = min(core.provider_stats()["products_ms"] for _ in range(3) if core.block_products(v, want=(True, False, False)) is not None)
print(json.dumps({"n": n, "me": me, "mi": mi, "block_products_ms": best["products_ms"], "bytes_moved": best["products_bytes"],
                  "GB_per_s": best["products_bytes"] / best["products_ms"] / 1e6,
                  "GB_per_s_counting_Q_twice_as_rounds_2_3_did": 8.0 * (n * n + n * me + n * mi) / best["products_ms"] / 1e6,
                  "Qv_only_ms": only, "Qv_only_GB_per_s": 4.0 * n * n / only / 1e6, "max_rel_err_vs_torch": err,
                  "same_bits_twice": bool(torch.equal(q, qo))}))
