#!/bin/bash
# round 5, first GPU call: the new batched / ADVICE tests, then the default bench line (new legs, CPU baseline at N = 32768)
set -u
O=gpurun_out/r05a; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
( timeout 1500 python -m pytest tests/test_gpu_batched.py tests/test_gpu_pivoting.py tests/test_gpu_ipm.py tests/test_gpu_merit.py tests/test_gpu_qp.py -m gpu -q -x --durations=5 > $O/pytest_a.log 2>&1; echo "pytest rc=$?" >> $O/pytest_a.log )
tail -15 $O/pytest_a.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 600 $O/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05a/bench_default.json"))
print("ms/step", d["ms_per_step"], "value", d["value"], "roof", d["roofline"]["frac"], "panel", d["phases_ms_per_step"])
for k in ("config2", "config3", "config5", "config4"):
    v = d.get(k, {})
    print(k, {kk: v.get(kk) for kk in ("ms_per_step", "ms_per_batch_step", "value", "backward_error", "chain_share_of_step")}, v.get("forms"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["legs"], d["cpu_baseline"]["cores"])
PY
