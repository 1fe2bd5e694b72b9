#!/bin/bash
set -u
O=gpurun_out/r02r; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_symmetric.py tests/test_gpu_parity.py tests/test_gpu_condensed.py tests/test_gpu_pivoting.py tests/test_gpu_ipm.py tests/test_gpu_qp.py tests/test_gpu_host_abi.py -m gpu -q > $O/pytest_a.log 2>&1; tail -8 $O/pytest_a.log
bash tools/r02_run_q8.sh "--nvar 2048 --neq 0 --nineq 2048 --steps 20 --warmup 3" "" "--opt fast_on_main=0"
bash tools/r02_run_q8.sh "--steps 5 --warmup 2" "" "--opt fast_on_main=0"
python - <<'PY'
import json
for f in ("ab_1_1","ab_2_1"):
    d=json.load(open("gpurun_out/r02q/%s.json"%f)); print(f, d["ms_per_step"], d["phases_ms_per_step"], d["backward_error"])
PY
