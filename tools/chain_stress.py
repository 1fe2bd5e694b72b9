"""Stress of k_tile_chain's hand-overs (kernels_chain.hpp) under UNEVEN load: random shapes / panel widths / group sizes, every step's
direction compared bit for bit with the launch-per-tile schedule, while a second stream keeps a varying number of compute units busy
with matrix products (timing of the chain's workgroups shifts from step to step); the workspace of every handle is poisoned with NaN.
usage: python tools/chain_stress.py [seconds] [seed]"""
import os
import sys
import time
os.environ["PYIPM_POISON_WORKSPACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyipm_amd.newton import NewtonCore, NewtonError
from pyipm_amd.problems import make_qp

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
side = torch.cuda.Stream()
a = torch.randn(2048, 2048, dtype=torch.float64, device="cuda")
t_end = time.time() + budget
n_steps = n_bad = n_shapes = 0
tally = {}
while time.time() < t_end:
    nb = int(rng.choice([128, 256, 256, 512]))
    n = int(rng.integers(300, 5000)); me = int(rng.integers(0, 1200)); mi = int(rng.integers(0, 2500))
    qp = make_qp(n, me, mi, int(rng.integers(0, 1000)))
    base = {"tail_group": int(rng.choice([1, 2, 4, 8])), "group": int(rng.choice([2, 4, 8] if nb <= 256 else [2, 4])), "skip_zeros": int(rng.integers(0, 2))}
    ref = None
    for variant in ({"tile_chain": 0}, {"tile_chain": 2}, {"tile_chain": 1, "chain_lds_kb": int(rng.choice([0, 100]))},
                    {"tile_chain": 2, "chain_whole": 0}, {"tile_chain": 2, "chain_cpy": int(rng.choice([1, 2, 3, 5, 9]))}):
        core = NewtonCore(n, me, mi, device=0, nb=nb)
        core.set_option("expert", 1)
        for k, v in {**base, **variant}.items():
            core.set_option(k, v)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        for rep in range(3):
            load = int(rng.integers(0, 4))
            with torch.cuda.stream(side):
                for _ in range(load):
                    b = a @ a
            try:
                dz, st = core.step(0.0, 0.0)
            except NewtonError as e:
                print("FAIL", (n, me, mi, nb), base, variant, rep, str(e)[:70], flush=True); n_bad += 1
                tally[str(sorted(variant))] = tally.get(str(sorted(variant)), 0) + 1
                if n_bad <= 6:
                    stt = core.kkt_storage()                    # (ncols, Npad): row j = column j of the lower triangle
                    bad = torch.isnan(stt)
                    cols = torch.nonzero(bad.any(dim=1)).flatten()
                    if cols.numel():
                        c0 = int(cols[0]); rows = torch.nonzero(bad[c0]).flatten()
                        print("   first column with NaN: %d (tile %d, panel %d); its NaN rows: %d .. %d (%d of them); columns with NaN: %d; Npad %d"
                              % (c0, c0 // 64, c0 // nb, int(rows[0]), int(rows[-1]), rows.numel(), cols.numel(), stt.shape[1]), flush=True)
                        c1 = int(cols[min(1, cols.numel() - 1)])
                        # tiles (row tile, col tile) of the first 6 NaN tiles
                        tl = set()
                        for c in cols[:400].tolist():
                            for r in torch.nonzero(bad[c]).flatten()[:4000:64].tolist():
                                tl.add((r // 64, c // 64))
                        print("   NaN tiles (row tile, col tile), first 12 by column:", sorted(tl, key=lambda x: (x[1], x[0]))[:12], flush=True)
                continue
            n_steps += 1
            if ref is None:
                ref = dz.clone()
            elif not torch.equal(dz, ref):
                print("BITS", (n, me, mi, nb), base, variant, rep, float((dz - ref).abs().max()), flush=True); n_bad += 1
                tally["BITS " + str(sorted(variant))] = tally.get("BITS " + str(sorted(variant)), 0) + 1
        core.close()
    n_shapes += 1
torch.cuda.synchronize()
print("chain_stress: %d shapes, %d steps, %d bad" % (n_shapes, n_steps, n_bad), tally)
