#!/bin/bash
# Bench lines for the other BASELINE.json configurations (1 GPU).  Output: gpurun_out/cfg_*.json
set -u
mkdir -p gpurun_out
B="timeout 900 python bench.py --no-cpu-baseline "
$B --nvar 2048 --neq 0 --nineq 2048 --steps 20 --warmup 3 > gpurun_out/cfg2_full.json 2> gpurun_out/cfg2_full.err
$B --nvar 2048 --neq 0 --nineq 2048 --steps 20 --warmup 3 --opt condensed=1 > gpurun_out/cfg2_cond.json 2> gpurun_out/cfg2_cond.err
$B --nvar 16384 --neq 8192 --nineq 8192 --steps 3 --warmup 1 > gpurun_out/cfg3_full.json 2> gpurun_out/cfg3_full.err
$B --nvar 16384 --neq 8192 --nineq 8192 --steps 3 --warmup 1 --opt condensed=1 > gpurun_out/cfg3_cond.json 2> gpurun_out/cfg3_cond.err
$B --steps 3 --warmup 1 --force-dist --nb 1024 > gpurun_out/metric_forcedist.json 2> gpurun_out/metric_forcedist.err
$B --steps 3 --warmup 1 --force-dist --nb 512 > gpurun_out/metric_forcedist_nb512.json 2> /dev/null
$B --steps 3 --warmup 1 --force-dist --nb 256 > gpurun_out/metric_forcedist_nb256.json 2> /dev/null
$B --steps 3 --warmup 1 --opt condensed=1 > gpurun_out/metric_condensed.json 2> /dev/null
$B --nvar 65536 --neq 0 --nineq 32768 --steps 1 --warmup 1 > gpurun_out/cfg4_1gpu.json 2> gpurun_out/cfg4_1gpu.err
$B --nvar 65536 --neq 0 --nineq 32768 --steps 1 --warmup 1 --opt condensed=1 > gpurun_out/cfg4_1gpu_cond.json 2> gpurun_out/cfg4_1gpu_cond.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/cfg*.json") + glob.glob("gpurun_out/metric_*.json")):
    try:
        d = json.load(open(f))
        print(f, "N=%d" % d["config"]["kkt_dim"], "%.4f steps/s" % d["value"], "%.2f ms" % d["ms_per_step"],
              "upd %.1f TF/s" % d["roofline"]["achieved"], "step %.1f TF/s" % d["step_tflops"],
              "berr", d.get("backward_error"), d["inertia"]["n_neg"], d["inertia"]["expected"], d["inertia"]["n_zero"])
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-600:])
PY
