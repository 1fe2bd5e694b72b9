#!/bin/bash
set -u
O=gpurun_out/r02g; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -8 $O/pytest.log
B="timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 2"
$B > $O/bench_base.json 2> $O/bench_base.err
$B --opt bwd_fused=0 > $O/bench_bwd0.json 2> $O/bench_bwd0.err
C2="--nvar 2048 --neq 0 --nineq 2048 --steps 20 --warmup 3"
$B $C2 > $O/cfg2_base.json 2> $O/cfg2_base.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02g/*.json")):
    try:
        d = json.load(open(f)); p = d["phases_ms_per_step"]
        print(f.split("/")[-1], "%.2f ms" % d["ms_per_step"], "upd %.1f TF/s" % d["roofline"]["achieved"],
              "panel %.2f trailing %.2f solve %.2f asm %.2f" % (p["panel(tile+scale+in-panel)"], p["trailing"], p["solve"], p["assemble"]),
              "berr", d.get("backward_error"), d["hbm_bound_kernels"].get("assemble_K1", {}).get("GB_per_s"))
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-600:])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find $O/prof -name "*.db" | head -1) $O/kernel_stats.txt 2>/dev/null | grep "k_bwd\|k_panel_inv\|k_assemble\|k_fwd\|k_update<128, true"
