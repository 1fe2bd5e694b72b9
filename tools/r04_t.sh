#!/bin/bash
set -u
O=gpurun_out/r04t; mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_symmetric.py tests/test_gpu_pivoting.py -q -x > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
for o in "" "--opt asm_tri=0" "" "--opt asm_tri=0"; do timeout 600 python bench.py --no-cpu-baseline --config4 off --steps 8 --warmup 2 $o 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$o', round(d['ms_per_step'],2), 'assemble', round(d['phases_ms_per_step']['assemble'],3), round(d['hbm_bound_kernels']['assemble_K1']['achieved']), 'GB/s')"; done
