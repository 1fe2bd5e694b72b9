#!/bin/bash
set -u
O=gpurun_out/r02h; mkdir -p $O
B="timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 2"
$B > $O/base.json 2> $O/base.err
$B --opt cu_split=-1 > $O/ctrl_allones.json 2> $O/ctrl_allones.err
$B --opt cu_split=8 --opt cu_side=0 > $O/s8_sideany.json 2> $O/s8_sideany.err
$B --opt cu_split=8 > $O/s8.json 2> $O/s8.err
$B --opt cu_split=16 > $O/s16.json 2> $O/s16.err
$B --opt cu_split=16 --opt cu_rows=8192 > $O/s16_r8k.json 2> $O/s16_r8k.err
$B --opt cu_split=16 --opt cu_rows=16384 > $O/s16_r16k.json 2> $O/s16_r16k.err
$B --opt cu_split=32 > $O/s32.json 2> $O/s32.err
$B --opt cu_split=24 --opt cu_rows=10240 > $O/s24_r10k.json 2> $O/s24_r10k.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02h/*.json")):
    try:
        d = json.load(open(f)); p = d["phases_ms_per_step"]
        print(f.split("/")[-1], "%.2f ms" % d["ms_per_step"], "upd %.1f TF/s" % d["roofline"]["achieved"],
              "panel %.2f trailing %.2f solve %.2f asm %.2f" % (p["panel(tile+scale+in-panel)"], p["trailing"], p["solve"], p["assemble"]),
              "berr", d.get("backward_error"))
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-600:])
PY
