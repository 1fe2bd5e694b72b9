#!/bin/bash
set -u
O=gpurun_out/r04k; mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_qp.py tests/test_gpu_dist.py tests/test_gpu_pivoting.py tests/test_gpu_parity.py tests/test_gpu_condensed.py tests/test_gpu_host_abi.py tests/test_gpu_merit.py -q --durations=5 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -8 $O/tests.txt
timeout 600 python tools/qp_solve.py > $O/qp_solve_full.json 2> $O/qp_solve_full.err
python -c "
import json; d=json.loads(open('$O/qp_solve_full.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('iterations','solve_seconds','newton_seconds','search_seconds','newton_seconds_per_factorisation')}); print(d['provider'])"
