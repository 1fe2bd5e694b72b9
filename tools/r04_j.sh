#!/bin/bash
set -u
O=gpurun_out/r04j; mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
PYIPM_SETUP_TRACE=1 timeout 300 python tools/first_call.py > $O/first_call.json 2> $O/first_call.err; grep pyipm $O/first_call.err
python -c "
import json; d=json.load(open('$O/first_call.json'))
for c in d['calls'][:2]: print({k: round(v,2) for k,v in c.items() if k.endswith('_ms')})"
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.txt 2>&1; echo "gpu suite rc=$?"; tail -14 $O/pytest_gpu.txt
timeout 600 python tools/qp_solve.py > $O/qp_solve_full.json 2> $O/qp_solve_full.err
python -c "
import json; d=json.loads(open('$O/qp_solve_full.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('iterations','factorisations','solve_seconds','newton_seconds','search_seconds','newton_seconds_per_factorisation','merit_ray_launches','rcond_estimates')}); print([round(x*1e3,1) for x in d['newton_seconds_each']])"
timeout 600 python bench.py --no-cpu-baseline --config4 off --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],2), d['phases_ms_per_step'])"
