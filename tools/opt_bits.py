"""Bitwise check of schedule options at a mid size where every phase of the schedule occurs: the direction with each option
set against the default's.  usage: python tools/opt_bits.py name=value [name=value ...] (each argument one variant)"""
import os
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_qp_device
from pyipm_amd.newton import NewtonCore

dev = torch.device("cuda", 0)
for shape in ((16384, 4096, 6144), (9000, 1000, 4000)):
    n, me, mi = shape
    qp = make_qp_device(n, me, mi, 1, dev)
    ref = None
    for variant in [""] + sys.argv[1:]:
        core = NewtonCore(n, me, mi, device=0)
        for kv in variant.split(","):
            if kv:
                k, v = kv.split("=")
                core.set_option(k, float(v))
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        outs = [core.step(0.0, 0.0)[0].clone() for _ in range(3)]
        if ref is None:
            ref = outs[0]
        print(shape, repr(variant), "same bits as default:", all(torch.equal(o, ref) for o in outs), flush=True)
        core.close()
    del qp
    torch.cuda.empty_cache()
