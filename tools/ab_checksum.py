#!/usr/bin/env python
"""A/B helper: bitwise fingerprints of the Newton direction and the factor statistics on a fixed set of
systems (convex QPs, the reference traces' indefinite / rank-deficient steps).  Run before and after a
kernel change that is meant to keep the arithmetic identical and diff the output."""
import glob, hashlib, os, sys
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyipm_amd.newton import NewtonCore
from pyipm_amd.problems import make_qp

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def fp(t):
    return hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:16]


for shape in [(24, 8, 16, 0), (96, 32, 48, 4), (300, 100, 150, 7), (700, 200, 300, 8), (1500, 300, 500, 11), (2048, 0, 2048, 0)]:
    n, me, mi, seed = shape
    qp = make_qp(n, me, mi, seed)
    for cond in (0, 1):
        if cond and not mi:
            continue
        core = NewtonCore(n, me, mi, device=0)
        core.set_option("condensed", cond)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]); core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        dz, st = core.step(0.0, 0.0)
        print("qp", shape, "cond", cond, fp(dz), st["n_neg"], st["n_zero"], st["n_2x2"], "%.17g %.17g %.17g" % (st["d_min"], st["d_max"], st["growth"]))
for f in sorted(glob.glob(os.path.join(GOLD, "step_*.npz"))):
    d = np.load(f)
    if "Q" not in d:
        continue
    n, me, mi = int(d["nvar"]), int(d["neq"]), int(d["nineq"])
    core = NewtonCore(n, me, mi, device=0)
    core.stage_blocks(d["Q"], d["A"].T.copy(), d["G"].T.copy())
    core.stage_vectors(d["Q"] @ d["x"] + d["c"], d["A"] @ d["x"] - 0.1, d["G"] @ d["x"] + 1.0, d["s"], d["lda"], mu=float(d["mu"]))
    for delta in (0.0, float(d["delta_out"])):
        core.residual(); core.assemble(delta, 0.0); st = core.factor()
        dz = core.solve()
        print(os.path.basename(f), delta, fp(dz), st["n_neg"], st["n_zero"], st["n_2x2"], "%.17g %.17g" % (st["d_min"], st["d_max"]))
# random symmetric indefinite d2L (exercises the 2x2 / off-diagonal pivot branches)
rng = np.random.default_rng(5)
for n in (64, 200, 512):
    M = rng.standard_normal((n, n)); H = (M + M.T) / 2
    core = NewtonCore(n, 0, 0, device=0)
    core.stage_blocks(H, None, None); core.stage_vectors(rng.standard_normal(n), None, None, None, None)
    dz, st = core.step(0.0, 0.0)
    print("indef", n, fp(dz), st["n_neg"], st["n_zero"], st["n_2x2"], "%.17g %.17g %.17g" % (st["d_min"], st["d_max"], st["growth"]))
