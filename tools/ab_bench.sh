#!/bin/bash
# A/B on one box (boxes differ by +-3 %): every argument is one set of bench.py options; each variant runs three times,
# interleaved.  Prints ms/step (exposed panel ms) per run.  usage: tools/ab_bench.sh "" "--opt group_chain=0" ...
set -u
O=gpurun_out/r02q; mkdir -p $O
B="timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 2"
for rep in 1 2 3; do
  i=0
  for o in "$@"; do
    i=$((i+1))
    $B $o > $O/ab_${i}_$rep.json 2> $O/ab_${i}_$rep.err
  done
done
python - "$@" <<'PY'
import json, sys, glob
opts = sys.argv[1:]
for i, o in enumerate(opts, 1):
    v = []
    for f in sorted(glob.glob("gpurun_out/r02q/ab_%d_*.json" % i)):
        try: d = json.load(open(f)); v.append((d["ms_per_step"], d["phases_ms_per_step"]["panel(tile+scale+in-panel)"]))
        except Exception as e: v.append(("FAIL", str(e)))
    print(repr(o), " ".join("%.2f(%.2f)" % x if x[0] != "FAIL" else "FAIL" for x in v))
PY
