"""Timeline of k_tile_chain launches (100 MHz device clock): per step of the chain workgroup -- wait for the row's units, operands
staged, scaling product, diagonal update, inversion, publish -- and when the units of row t finished the stage the chain waits for.
usage: python tools/chain_clock.py N ME MI [launch_index ...] [name=value ...]"""
import os
import sys
os.environ.setdefault("PYIPM_EXPERT", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import make_qp_device
from pyipm_amd.newton import NewtonCore

args = [a for a in sys.argv[1:] if "=" not in a]
opts = [a for a in sys.argv[1:] if "=" in a]
n, me, mi = (int(v) for v in args[:3])
which = [int(v) for v in args[3:]] or [0, 1]
W = 256 + 64 * 4 * 32
qp = make_qp_device(n, me, mi, 0, torch.device("cuda", 0))
core = NewtonCore(n, me, mi, device=0)
core.set_option("tile_chain", 2)
for kv in opts:
    k, v = kv.split("=")
    core.set_option(k, float(v))
core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
for _ in range(3):
    core.step(0.0, 0.0)
buf = torch.zeros(64 * W, dtype=torch.int64, device="cuda")
core.set_option("debug_chain_ptr", float(buf.data_ptr()))
core.step(0.0, 0.0)
torch.cuda.synchronize()
core.set_option("debug_chain_ptr", 0.0)
b = buf.cpu().numpy().reshape(64, W)
for li in which:
    d = b[li]
    steps = [t for t in range(32) if d[8 * t + 6] != 0]
    if not steps:
        continue
    t00 = d[8 * steps[0]]
    print("launch %d: steps %d..%d (us; 100 MHz clock)" % (li, steps[0], steps[-1]))
    print("  t   start   wait  stage  scale   diag invert publish | step   units of row t: last stage done at (rel. to chain's wait begin)")
    for t in steps:
        s = d[8 * t: 8 * t + 7].astype(np.float64) / 100.0
        if t == 0 or d[8 * t + 1] == 0:
            print(" %2d %7.2f      -      -      -      - %6.2f %6.2f  | %5.2f" % (t, (d[8 * t] - t00) / 100.0, s[5] - s[0], s[6] - s[5], s[6] - s[0]))
            continue
        row = []
        for y in range(4):
            w = d[256 + 64 * (4 * t + y) + 2 * (t - 2) + 1] if t >= 2 else 0
            if w:
                row.append("%+.2f" % ((w - d[8 * t]) / 100.0))
        print(" %2d %7.2f %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f  | %5.2f   %s" % (
            t, (d[8 * t] - t00) / 100.0, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], s[6] - s[5], s[6] - s[0], " ".join(row)))
    # the units: duration of each stage of the furthest row
    r = max(rr for rr in range(32) if any(d[256 + 64 * (4 * rr + y) + 1] for y in range(4)) or rr == 0)
    for y in range(4):
        st = [(tp, d[256 + 64 * (4 * r + y) + 2 * tp], d[256 + 64 * (4 * r + y) + 2 * tp + 1]) for tp in range(31)]
        st = [x for x in st if x[2]]
        if st:
            print("  unit (%d, %d): stage: begin(after waits) -> published, us since launch start: " % (r, y) +
                  "  ".join("%d: %.1f -> %.1f" % (tp, (a - t00) / 100.0, (e - t00) / 100.0) for tp, a, e in st))
core.close()
