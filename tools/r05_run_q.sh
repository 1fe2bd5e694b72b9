#!/bin/bash
# config 2 kernel timelines under option variants: when does the bulk update of the first x group start?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in "default:" "hwq8:" "hwq2:"; do
  name=${v%%:*}; args=${v#*:}
  if [ $name = hwq8 ]; then export GPU_MAX_HW_QUEUES=8; fi
  if [ $name = hwq2 ]; then export GPU_MAX_HW_QUEUES=2; fi
  OUTDIR=r05q_$name bash tools/config2_timeline.sh --config4 off --configs off --no-clock $args > /dev/null 2>&1
  python - $name <<'PY'
import sys
name = sys.argv[1]
rows = []
for ln in open("gpurun_out/r05q_%s/c2_timeline.txt" % name):
    p = ln.split()
    if len(p) < 5 or p[0].startswith("#") or p[0] == "start_us": continue
    rows.append((float(p[0]), float(p[1]), float(p[2]), p[3], " ".join(p[4:])))
ia = max(i for i, r in enumerate(rows) if "k_assemble" in r[4])
rows = rows[ia:]; t0 = rows[0][0]
ts = [r for r in rows if "k_tile_step" in r[4]]
print(name, "queues", sorted(set(r[3] for r in rows)), "tile steps %d, first %.0f last end %.0f; k_update starts/durs: %s; bwd_sweep at %s" % (
    len(ts), ts[0][0] - t0, ts[-1][0] + ts[-1][1] - t0,
    [(round(r[0] - t0), round(r[1])) for r in rows if "k_update" in r[4]],
    [(round(r[0] - t0), round(r[1])) for r in rows if "k_bwd_sweep" in r[4]]))
PY
done
