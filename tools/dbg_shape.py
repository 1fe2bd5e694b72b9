"""Which option makes the direction of a given shape differ?  Backward error from the blocks and the relative difference to the
default for a list of option variants.  usage: python tools/dbg_shape.py n me mi [variant ...]  (variant: a=1,b=2)"""
import os
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_qp_device
from pyipm_amd.newton import NewtonCore

n, me, mi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
qp = make_qp_device(n, me, mi, 5, dev)
ref = None
for variant in [""] + sys.argv[4:]:
    core = NewtonCore(n, me, mi, device=0)
    core.set_option("profile", 1)
    for kv in variant.split(","):
        if kv:
            k, v = kv.split("=")
            core.set_option(k, float(v))
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    dz, st = core.step(0.0, 0.0)
    g = core.residual()
    raw = dz.clone(); raw[n + mi:] *= -1.0
    berr = float((core.matvec(raw) - g).norm() / g.norm())
    if ref is None:
        ref = dz.clone()
    inst = core.trailing_instances()
    print(repr(variant), "berr %.2e" % berr, "rel.diff to default %.2e" % float((dz - ref).norm() / ref.norm()),
          "neg %d zero %d" % (st["n_neg"], st["n_zero"]), "launches 128/256: %d/%d" % (inst[128]["launches"], inst[256]["launches"]), flush=True)
    core.close()
