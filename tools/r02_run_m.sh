#!/bin/bash
set -u
O=gpurun_out/r02m; mkdir -p $O
python tools/tile_clock.py 2>&1 | grep "last tile"
( timeout 2400 python -m pytest tests -m gpu -q -x --durations=3 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -6 $O/pytest.log
B="timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 2"
$B > $O/bench_base.json 2> $O/bench_base.err
C2="--nvar 2048 --neq 0 --nineq 2048 --steps 20 --warmup 3"
$B $C2 > $O/cfg2_base.json 2> $O/cfg2_base.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02m/*.json")):
    d = json.load(open(f)); p = d["phases_ms_per_step"]
    print(f.split("/")[-1], "%.2f ms" % d["ms_per_step"], "upd %.1f TF/s" % d["roofline"]["achieved"],
          "panel %.2f trailing %.2f solve %.2f asm %.2f" % (p["panel(tile+scale+in-panel)"], p["trailing"], p["solve"], p["assemble"]), "berr", d.get("backward_error"))
PY
