"""Where is the step exposed?  From a metric_timeline.txt (tools/step_timeline.sh): the intervals of the last step during which
NO bulk update launch (k_update<*, true, 8> on the main queue) runs, with what ran inside them.
usage: python tools/exposed_gaps.py gpurun_out/<dir>/metric_timeline.txt"""
import sys
rows = []
for ln in open(sys.argv[1]):
    p = ln.split()
    if len(p) < 5 or p[0].startswith("#") or p[0] == "start_us":
        continue
    rows.append((float(p[0]), float(p[1]), p[3], " ".join(p[4:])))
# last step: from the last k_assemble on
ia = max(i for i, r in enumerate(rows) if "k_assemble" in r[3])
rows = rows[ia:]
t0 = rows[0][0]
bulk = sorted((r[0], r[0] + r[1]) for r in rows if "k_update<" in r[3] and "true, 8>" in r[3])
end = max(r[0] + r[1] for r in rows)
gaps, cur = [], rows[0][0]
for a, b in bulk:
    if a > cur + 20:
        gaps.append((cur, a))
    cur = max(cur, b)
if end > cur + 20:
    gaps.append((cur, end))
print("step %.1f ms, %d bulk launches covering %.1f ms, exposed %.1f ms" % ((end - t0) / 1e3, len(bulk), sum(b - a for a, b in bulk) / 1e3,
                                                                            sum(b - a for a, b in gaps) / 1e3))
for a, b in gaps:
    inside = {}
    for r in rows:
        ov = min(b, r[0] + r[1]) - max(a, r[0])
        if ov > 0:
            k = r[3].split("<")[0] if "k_update" not in r[3] else r[3]
            inside[k] = inside.get(k, [0, 0.0]); inside[k][0] += 1; inside[k][1] += ov
    top = sorted(inside.items(), key=lambda kv: -kv[1][1])[:5]
    print("  gap at %8.1f us, %7.1f us: %s" % (a - t0, b - a, ", ".join("%s x%d %.0fus" % (k, v[0], v[1]) for k, v in top)))
