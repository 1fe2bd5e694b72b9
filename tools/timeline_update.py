#!/usr/bin/env python
"""Per-block timeline of one isolated trailing update (diagnostics)."""
import os, sys
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyipm_amd.newton import NewtonCore
from bench import make_qp_device
n, me, mi, nb = 16384, 4096, 6144, int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
qp = make_qp_device(n, me, mi, 0, dev)
core = NewtonCore(n, me, mi, device=0, nb=nb)
core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]); core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
for kv in sys.argv[2:]:
    k, v = kv.split("="); core.set_option(k, float(v))
core.assemble(0.0, 0.0); core.factor_begin(); core.factor_panel(0)
core.trailing_update(0); torch.cuda.synchronize()
nblk = (core.Npad // 128) ** 2 + 4096
buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
core.set_option("debug_timeline_ptr", float(buf.data_ptr()))
core.trailing_update(0); torch.cuda.synchronize()
core.set_option("debug_timeline_ptr", 0.0)
d = buf.cpu().numpy().reshape(-1, 8)
d = d[d[:, 0] != 0]
t0 = d[:, 0].min()
T = (d[:, :4] - t0) * 0.01   # us (100 MHz)
pro, loop, epi = T[:, 1] - T[:, 0], T[:, 2] - T[:, 1], T[:, 3] - T[:, 2]
print("blocks timed:", len(d), " kernel span %.1f us" % (T[:, 3].max()))
for name, x in (("prologue(C+stage0 load)", pro), ("main loop", loop), ("epilogue(stores issued)", epi), ("total", T[:, 3] - T[:, 0])):
    print("%-26s mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f us" % (name, x.mean(), *np.percentile(x, [10, 50, 90])))
hw = d[:, 4] & 0xffff; xcc = d[:, 5]
cyc = d[:, 4] >> 16
mhz = cyc / np.maximum(loop, 1e-9)
print("shader clock over the main loop (clock64 ticks / wall time): mean %.0f MHz  p10 %.0f  p50 %.0f  p90 %.0f" % (mhz.mean(), *np.percentile(mhz, [10, 50, 90])))
wave_id = hw & 0xF; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
drain = d[:, 6] * 0.01
print("store drain (vmcnt(0) after last store issue): mean %.2f p50 %.2f p90 %.2f us" % (drain.mean(), *np.percentile(drain, [50, 90])))
print("wave_id values:", np.unique(wave_id), " simd:", np.unique(simd), " cu:", np.unique(cu), " sh:", np.unique(sh), " se:", np.unique(se), " xcc:", np.unique(xcc))
key = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + wave_id
gaps = []
for k in np.unique(key):
    idx = np.where(key == k)[0]
    idx = idx[np.argsort(T[idx, 0])]
    gaps.extend((T[idx[1:], 0] - T[idx[:-1], 3]).tolist())
gaps = np.array(gaps)
print("slots:", len(np.unique(key)), " gap end->next start on same slot: mean %.2f p50 %.2f p90 %.2f us" % (gaps.mean(), *np.percentile(gaps, [50, 90])))
# phase relation of CU-mates: at each block's prologue midpoint, is the mate in its main loop?
cukey = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10
inloop = 0; tot = 0
for k in np.unique(cukey)[:64]:
    idx = np.where(cukey == k)[0]
    for i in idx:
        mid = 0.5 * (T[i, 0] + T[i, 1])
        others = idx[idx != i]
        tot += 1
        inloop += np.any((T[others, 1] <= mid) & (mid <= T[others, 2]))
print("prologue midpoints during which a CU-mate is in its main loop: %d / %d" % (inloop, tot))
first = np.argsort(T[:, 0])[:12]
for i in first: print("blk x=%d ct=%d start %.2f pro %.2f loop %.2f epi %.2f wave_id %d cu %d se %d xcc %d" % (d[i,6], d[i,7], T[i,0], pro[i], loop[i], epi[i], wave_id[i], cu[i], se[i], xcc[i]))
