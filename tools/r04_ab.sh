#!/bin/bash
# A/B on one box (boxes differ by +-3 %): every argument is one set of bench.py options; each variant runs $REPS times (default 3),
# interleaved.  Prints ms/step (exposed panel ms, all-bulk TF/s) per run and a bitwise fingerprint check of dz for the options.
# usage: tools/r04_ab.sh TAG "" "--opt ingroup_waves=8" ...
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
B="timeout 600 python bench.py --no-cpu-baseline --config4 off --steps ${STEPS:-6} --warmup 2 ${EXTRA:-}"
for rep in $(seq 1 ${REPS:-3}); do
  i=0
  for o in "$@"; do
    i=$((i+1))
    $B $o > $O/ab_${i}_$rep.json 2> $O/ab_${i}_$rep.err
  done
done
python - "$O" "$@" <<'PY'
import json, sys, glob
O = sys.argv[1]; opts = sys.argv[2:]
for i, o in enumerate(opts, 1):
    v = []
    for f in sorted(glob.glob("%s/ab_%d_*.json" % (O, i))):
        try:
            d = json.load(open(f))
            v.append("%.2f(p%.2f b%.1f berr%.0e)" % (d["ms_per_step"], d["phases_ms_per_step"]["panel(tile+scale+in-panel)"],
                                                 d["roofline"]["all_bulk_launches"]["achieved"], d["backward_error"]))
        except Exception as e:
            v.append("FAIL")
    print(repr(o), " ".join(v))
PY
