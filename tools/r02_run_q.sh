#!/bin/bash
# stepped panel schedule: parity first, then the chain-bound and the headline configurations with and without it
set -u
O=gpurun_out/r02q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_symmetric.py tests/test_gpu_parity.py tests/test_gpu_condensed.py tests/test_gpu_pivoting.py -m gpu -x -q > $O/pytest_a.log 2>&1; tail -4 $O/pytest_a.log
B="timeout 600 python bench.py --no-cpu-baseline"
$B --nvar 2048 --neq 0 --nineq 2048 --steps 20 --warmup 3 > $O/c2_new.json 2> $O/c2_new.err
$B --nvar 2048 --neq 0 --nineq 2048 --steps 20 --warmup 3 --opt group_chain=0 > $O/c2_old.json 2> $O/c2_old.err
$B --steps 5 --warmup 2 > $O/m_new.json 2> $O/m_new.err
$B --steps 5 --warmup 2 --opt group_chain=0 > $O/m_old.json 2> $O/m_old.err
python - <<'PY'
import json
for f in ("c2_new","c2_old","m_new","m_old"):
    try:
        d=json.load(open("gpurun_out/r02q/%s.json"%f)); print(f, "%.3f ms"%d["ms_per_step"], d["phases_ms_per_step"], "upd %.1f TF"%d["roofline"]["achieved"], d["inertia"], d["backward_error"])
    except Exception as e:
        print(f,"FAILED",e, open("gpurun_out/r02q/%s.err"%f).read()[-800:])
PY
timeout 300 python tools/tile_clock.py 2>&1 | tail -3
