import os, sys
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyipm_amd.newton import NewtonCore
from pyipm_amd.problems import make_qp
n, me, mi = 1024, 256, 384
qp = make_qp(n, me, mi, 1)
core = NewtonCore(n, me, mi, device=0)
core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]); core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
core.set_option("lookahead", 0)
buf = torch.zeros(256, dtype=torch.int64, device="cuda")
if len(sys.argv) > 1:
    core.set_option("tile_blocked", float(sys.argv[1]))
for rep in range(3):
    core.assemble(0.0, 0.0); core.factor()
core.set_option("debug_timeline_ptr", float(buf.data_ptr()))
core.assemble(0.0, 0.0); core.factor(); torch.cuda.synchronize()
core.set_option("debug_timeline_ptr", 0.0)
c, w, k = buf[:3].tolist()
print("last tile kernel: %d shader cycles, %d wall ticks (%.2f us) -> %.0f MHz, %.0f cycles per sweep" % (c, w, w * 0.01, c / (w * 0.01), c / 64.0))

b = buf.tolist()
if b[6]:
    print("blocked path: preamble %d cycles, sweep until %d, whole %d" % (b[4], b[5], c))
    names = ["loaded", "elim", "mfma", "commit", "issued", "U", "bar1", "-"]
    for kb in range(4):
        for w in range(4):
            t = [x - b[6] for x in b[8 + 32 * kb + 8 * w: 16 + 32 * kb + 8 * w]]
            print("  block %d wave %d: loads issued @%d, loaded @%d, elimination done @%d (+%d), U @%d, MFMA phase done @%d (+%d), past barrier @%d, committed @%d"
                  % (kb, w, t[4], t[0], t[1], t[1] - t[0], t[5], t[2], t[2] - t[1], t[6], t[3]))

if b[200]:
    t = [x - b[200] for x in b[200:205]]
    print("k_tile_step block 0 (last launch): operands staged @%d, scaling product done @%d, L stored @%d, diagonal tile updated + staged @%d cycles; inversion starts @%d (clock of its own stamp: %d after entry)"
          % (t[1], t[2], t[3], t[4], b[6] - b[200], b[6] - b[200]))
