#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace as a --stats style table (name, calls, total, avg, %)."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["%-90s %8s %14s %12s %12s %12s %7s" % ("KERNEL", "CALLS", "TOTAL_ns", "AVG_ns", "MIN_ns", "MAX_ns", "PCT")]
    for r in rows:
        lines.append("%-90s %8d %14d %12.0f %12d %12d %6.2f%%" % (r[0][:90], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
