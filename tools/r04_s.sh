#!/bin/bash
set -u
O=gpurun_out/r04s; mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 600 python tools/opt_bits.py asm_split=0 > $O/opt_bits.txt 2>&1; grep -v amdgpu $O/opt_bits.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "suite rc=$?"; tail -4 $O/pytest.txt
REPS=3 bash tools/r04_ab.sh r04s_ab "" "--opt asm_split=0"
timeout 600 python tools/qp_solve.py > $O/qp_solve_full.json 2> $O/qp.err
python -c "
import json; d=json.loads(open('$O/qp_solve_full.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('iterations','solve_seconds','newton_seconds','search_seconds','newton_seconds_per_factorisation','warm_up_seconds_in_constructor')}); print([round(x*1e3,1) for x in d['newton_seconds_each']])"
