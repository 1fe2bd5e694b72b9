"""Do the shapes of tests/test_gpu_tile_blocked.py::test_eight_wave_tile_step... reach the refined block solves and the single-sweep
hand-over?  Per shape: does block_refine = 0 change the bits (some tile is flagged), does tile_blocked = 0 (other rounding order)."""
import os, sys
os.environ.setdefault("PYIPM_EXPERT", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from pyipm_amd.newton import NewtonCore
from test_gpu_tile_blocked import _graded_qp
for shape in [(700, 100, 200, 1, 5.0), (1100, 0, 300, 2, 9.0), (520, 130, 0, 3, 3.0), (2100, 300, 500, 4, 6.0)]:
    n, me, mi, seed, dec = shape
    qp = _graded_qp(n, me, mi, seed, dec)
    res = {}
    for key, opts in (("default", {}), ("no block refinement", {"block_refine": 0}), ("single sweeps", {"tile_blocked": 0})):
        core = NewtonCore(n, me, mi, device=0)
        core.set_option("tile_waves", 9)                 # (eight waves everywhere)
        for k, v in opts.items():
            core.set_option(k, v)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]); core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        dz, st = core.step(0.0, 0.0)
        res[key] = dz.clone()
        core.close()
    print(shape, {k: (bool(torch.equal(v, res["default"])), float((v - res["default"]).norm() / res["default"].norm())) for k, v in res.items() if k != "default"})
