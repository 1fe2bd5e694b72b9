#!/bin/bash
set -u
O=gpurun_out/r02e; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q --durations=8 -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -12 $O/pytest.log
B="timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 2"
$B > $O/bench_base.json 2> $O/bench_base.err
$B --force-dist --nb 1024 > $O/bench_fdist_native_nb1024.json 2> $O/bench_fdist_native_nb1024.err
$B --force-dist --nb 512 --selfmsg > $O/bench_fdist_native_nb512_selfmsg.json 2> $O/bench_fdist_native_nb512_selfmsg.err
$B --nb 512 > $O/bench_nb512.json 2> $O/bench_nb512.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02e/*.json")):
    try:
        d = json.load(open(f)); p = d["phases_ms_per_step"]
        print(f.split("/")[-1], "%.2f ms" % d["ms_per_step"], "upd %.1f TF/s" % d["roofline"]["achieved"],
              "panel %.2f trailing %.2f solve %.2f asm %.2f" % (p["panel(tile+scale+in-panel)"], p["trailing"], p["solve"], p["assemble"]),
              "berr", d.get("backward_error"), d.get("dist_phases_per_step"))
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-600:])
PY
timeout 600 python tools/qp_solve.py > $O/qp_solve_full.json 2> $O/qp_solve_full.err; tail -c 900 $O/qp_solve_full.json
