#!/usr/bin/env python
"""Timeline of one step from a rocprofv3 rocpd (.db) kernel trace: every launch of the last `nlast` kernels in start order
(offset, duration, gap to the previous end on the same queue).  Usage: chain_timeline.py trace.db [nlast] [out.txt]"""
import sqlite3
import sys


def main(path, nlast=400, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = "%s, start, end%s" % (name_col, (", " + qcol) if qcol else "")
    rows = cur.execute("select %s from kernels order by start" % sel).fetchall()
    rows = rows[-nlast:]
    t0 = rows[0][1]
    last_end = {}
    lines = ["# columns: %s" % cols, "%10s %9s %9s %4s  %s" % ("start_us", "dur_us", "gap_us", "q", "kernel")]
    for r in rows:
        q = r[3] if qcol else 0
        gap = (r[1] - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = r[2]
        nm = r[0].replace("pyipm::", "").split("(")[0][:40]
        lines.append("%10.1f %9.2f %9.2f %4s  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, q, nm))
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    else:
        print(txt)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 400, sys.argv[3] if len(sys.argv) > 3 else None)
