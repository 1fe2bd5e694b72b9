#!/usr/bin/env python
"""Where does a Newton direction of the device QP loop spend its time beyond the factorisation?  From a rocprofv3 rocpd
kernel trace of tools/qp_solve.py: per iterate (k_assemble to k_assemble) the wall time, the time no kernel runs at all
(host round trips), and the kernels outside the factorisation window by name.
usage: python tools/qp_timeline.py trace.db [out.txt]"""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    rows = [(r[0].replace("pyipm::", "").split("(")[0][:48], r[1], r[2]) for r in rows]
    asm = [i for i, r in enumerate(rows) if r[0].startswith("void k_assemble") or r[0].startswith("k_assemble")]
    lines = []
    for a, b in zip(asm[:-1], asm[1:]):
        it = rows[a:b]
        t0, t1 = it[0][1], rows[b][1]
        # union of busy intervals
        busy, cur_e = 0, t0
        for _, s, e in it:
            if e > cur_e:
                busy += e - max(s, cur_e); cur_e = e
        # the factorisation window: up to the end of the last update launch of the iterate
        upd_end = max((e for n, s, e in it if "k_update" in n or "k_tile_step" in n or "k_panel_rest" in n), default=t0)
        outside = {}
        for n, s, e in it:
            if s >= upd_end:
                k = n.replace("void ", "")
                outside[k] = outside.get(k, [0, 0]); outside[k][0] += 1; outside[k][1] += e - s
        idle_after, cur_e = 0, upd_end
        for n, s, e in it:
            if s >= upd_end:
                if s > cur_e: idle_after += s - cur_e
                cur_e = max(cur_e, e)
        idle_after += max(0, t1 - cur_e)
        top = sorted(outside.items(), key=lambda kv: -kv[1][1])[:8]
        lines.append("iterate: wall %.2f ms, factor window %.2f, after it %.2f (idle %.2f): %s" % (
            (t1 - t0) / 1e6, (upd_end - t0) / 1e6, (t1 - upd_end) / 1e6, idle_after / 1e6,
            ", ".join("%s x%d %.2f" % (k, v[0], v[1] / 1e6) for k, v in top)))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
