"""Which schedule option is not bitwise-neutral?  Every (name, value) of tests/test_gpu_symmetric.py:SCHEDULE_SPACE alone against
the defaults, at the shapes of test_random_schedule_options_give_the_same_bits (per-panel sweeps, as there)."""
import os
os.environ.setdefault("PYIPM_EXPERT", "1")
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pyipm_amd.newton import NewtonCore
from pyipm_amd.problems import make_qp
from test_gpu_symmetric import SCHEDULE_SPACE

for shape, nb in (((3000, 700, 1200, 3), 256), ((1900, 300, 900, 9), 128), ((1000, 300, 900, 2), 256)):
    n, me, mi, seed = shape
    qp = make_qp(n, me, mi, seed)

    def run(opts):
        core = NewtonCore(n, me, mi, device=0, nb=nb)
        core.set_option("sweep_persist", 0)
        for k, v in opts.items():
            core.set_option(k, v)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        outs = [core.step(0.0, 0.0)[0].clone() for _ in range(2)]
        core.close()
        return outs

    ref = run({})[0]
    bad = []
    for name, vals in SCHEDULE_SPACE.items():
        for v in vals:
            try:
                outs = run({name: v})
                if not all(torch.equal(o, ref) for o in outs):
                    bad.append((name, v, float((outs[0] - ref).norm() / ref.norm())))
            except Exception as e:
                bad.append((name, v, "ERROR %s" % e))
    print(shape, nb, "not bitwise-neutral:", bad, flush=True)
