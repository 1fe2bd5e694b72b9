#!/bin/bash
set -u
O=gpurun_out/r05n; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYIPM_EXPERT=1
timeout 600 python bench.py --no-cpu-baseline --config4 off --configs off --no-clock --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('metric', round(d['ms_per_step'], 2), d['phases_ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --config4 off --configs off --no-clock --nvar 16384 --neq 8192 --nineq 8192 --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg3', round(d['ms_per_step'], 2), d['phases_ms_per_step'])"
timeout 600 python tools/qp_solve.py 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('qp', d['solve_seconds'], d['newton_seconds_per_factorisation'], [round(1e3 * x, 1) for x in d['newton_seconds_each']])"
timeout 1800 python -m pytest tests/test_gpu_pivoting.py tests/test_gpu_ipm.py tests/test_gpu_condensed.py tests/test_gpu_lbfgs.py tests/test_gpu_fuzz_backends.py -x -q > $O/pytest_sw2.log 2>&1; tail -3 $O/pytest_sw2.log
