#!/bin/bash
set -u
O=gpurun_out/r02i; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -8 $O/pytest.log
B="timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 2"
$B > $O/bench_base.json 2> $O/bench_base.err
$B --opt fuse_scale_update=0 > $O/bench_fsu0.json 2> $O/bench_fsu0.err
C2="--nvar 2048 --neq 0 --nineq 2048 --steps 20 --warmup 3"
$B $C2 > $O/cfg2_base.json 2> $O/cfg2_base.err
$B $C2 --opt fuse_scale_update=0 > $O/cfg2_fsu0.json 2> $O/cfg2_fsu0.err
$B --force-dist --nb 512 > $O/fdist512.json 2> $O/fdist512.err
python tools/bench_tile.py > $O/bench_tile.txt 2>&1; cat $O/bench_tile.txt | tail -1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02i/*.json")):
    try:
        d = json.load(open(f)); p = d["phases_ms_per_step"]
        print(f.split("/")[-1], "%.2f ms" % d["ms_per_step"], "upd %.1f TF/s" % d["roofline"]["achieved"],
              "panel %.2f trailing %.2f solve %.2f asm %.2f" % (p["panel(tile+scale+in-panel)"], p["trailing"], p["solve"], p["assemble"]),
              "berr", d.get("backward_error"))
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-600:])
PY
