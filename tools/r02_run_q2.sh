#!/bin/bash
set -u
O=gpurun_out/r02q; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_symmetric.py tests/test_gpu_parity.py tests/test_gpu_condensed.py tests/test_gpu_pivoting.py -m gpu -q > $O/pytest_a.log 2>&1; tail -15 $O/pytest_a.log
B="timeout 600 python bench.py --no-cpu-baseline"
$B --steps 2 --warmup 1 --opt pending32_rows=1000000000 > $O/m_p32.json 2> $O/m_p32.err
$B --steps 2 --warmup 1 --nvar 16384 --neq 0 --nineq 8192 > $O/m_me0.json 2> $O/m_me0.err
$B --steps 2 --warmup 1 --opt early_head=0 > $O/m_noearly.json 2> $O/m_noearly.err
$B --steps 2 --warmup 1 --opt skip_zeros=0 > $O/m_noskip.json 2> $O/m_noskip.err
python - <<'PY'
import json
for f in ("m_p32","m_me0","m_noearly","m_noskip"):
    try:
        d=json.load(open("gpurun_out/r02q/%s.json"%f)); print(f, "%.3f ms"%d["ms_per_step"], d["inertia"], d["backward_error"])
    except Exception as e:
        print(f,"FAILED",e, open("gpurun_out/r02q/%s.err"%f).read()[-800:])
PY
