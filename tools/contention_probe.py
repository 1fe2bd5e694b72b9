#!/usr/bin/env python
"""How much does a saturating bulk update (k_update) slow the tile chain down?  One handle loops trailing updates of a big
system on one stream while another handle factors a small, chain-bound system (N = 2048: 32 tile launches) on a second
stream.  Prints the chain's time per tile alone and beside the bulk kernel, and the bulk kernel's rate alone / beside.
usage: contention_probe.py [opt=value ...]   (options go to both handles)"""
import os, sys, time
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyipm_amd.newton import NewtonCore
from pyipm_amd.problems import make_qp
from bench import make_qp_device

dev = torch.device("cuda", 0)
opts = [kv.split("=") for kv in sys.argv[1:]]
big = make_qp_device(16384, 4096, 6144, 0, dev)
cb = NewtonCore(16384, 4096, 6144, device=0, nb=1024)
cb.stage_blocks(big["d2L"], big["Je"], big["Ji"]); cb.stage_vectors(big["df"], big["ce"], big["ci"], big["s"], big["lam"], mu=big["mu"])
cb.set_option("skip_zeros", 0)
small = make_qp(1024, 256, 384, 1)
cs = NewtonCore(1024, 256, 384, device=0)
cs.stage_blocks(small["d2L"], small["Je"], small["Ji"]); cs.stage_vectors(small["df"], small["ce"], small["ci"], small["s"], small["lam"], mu=small["mu"])
cs.set_option("lookahead", 0)
for k, v in opts:
    cb.set_option(k, float(v)); cs.set_option(k, float(v))
sb, ss = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
with torch.cuda.stream(sb):
    cb.assemble(0.0, 0.0); cb.factor_begin(); cb.factor_panel(0); cb.trailing_update(0)
torch.cuda.synchronize()
m = cb.Npad - 1024
flops = 2.0 * 1024 * (m * (m + 1) / 2)


def chain(reps=4):
    ts = []
    for _ in range(reps):
        with torch.cuda.stream(ss):
            cs.assemble(0.0, 0.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ss); cs.factor(); e1.record(ss)
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def bulk(n):
    with torch.cuda.stream(sb):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(sb)
        for _ in range(n):
            cb.trailing_update(0)
        e1.record(sb)
    return e0, e1


chain(2)
alone = chain()
e0, e1 = bulk(3); torch.cuda.synchronize()
print("chain alone: %.1f us per tile launch (N=2048: %d tiles); bulk alone: %.1f TF/s" % (1e3 * alone / (cs.Npad / 64), cs.Npad // 64, 3 * flops / e0.elapsed_time(e1) / 1e9))
e0, e1 = bulk(12)
time.sleep(0.01)
beside = chain(6)
torch.cuda.synchronize()
print("chain beside the bulk kernel: %.1f us per tile launch; bulk beside the chain: %.1f TF/s" % (1e3 * beside / (cs.Npad / 64), 12 * flops / e0.elapsed_time(e1) / 1e9))
