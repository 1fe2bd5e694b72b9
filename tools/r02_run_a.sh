#!/bin/bash
# round 2, GPU call A: full GPU suite + default bench + CU-mask experiment.  Output under gpurun_out/r02a/
set -u
O=gpurun_out/r02a; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ) 
tail -5 $O/pytest.log
./tools/ubench/cumask_probe > $O/cumask_probe.txt 2>&1; cat $O/cumask_probe.txt
B="timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 2"
$B > $O/bench_base.json 2> $O/bench_base.err
for k in 4 8 16 32; do for m in 0 1; do
  $B --opt reserve_cus=$k --opt reserve_mode=$m > $O/bench_res${k}_m${m}.json 2> $O/bench_res${k}_m${m}.err
done; done
C2="--nvar 2048 --neq 0 --nineq 2048 --steps 20 --warmup 3"
$B $C2 > $O/cfg2_base.json 2> $O/cfg2_base.err
$B $C2 --opt reserve_cus=8 --opt reserve_mode=1 > $O/cfg2_res8.json 2> $O/cfg2_res8.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02a/*.json")):
    try:
        d = json.load(open(f)); p = d["phases_ms_per_step"]
        print(f.split("/")[-1], "%.2f ms" % d["ms_per_step"], "upd %.1f TF/s" % d["roofline"]["achieved"],
              "panel %.2f trailing %.2f solve %.2f asm %.2f" % (p["panel(tile+scale+in-panel)"], p["trailing"], p["solve"], p["assemble"]),
              "berr", d.get("backward_error"))
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-400:])
PY
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench_default_full.json 2> $O/bench_default_full.err; tail -c 1500 $O/bench_default_full.json
