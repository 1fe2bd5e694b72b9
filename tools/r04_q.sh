#!/bin/bash
set -u
O=gpurun_out/r04q; mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 600 python tools/opt_bits.py asm_split=0 > $O/opt_bits.txt 2>&1; grep -v amdgpu $O/opt_bits.txt
timeout 1500 python -m pytest tests/test_gpu_pivoting.py tests/test_gpu_symmetric.py tests/test_gpu_parity.py tests/test_gpu_batched.py tests/test_gpu_ipm.py tests/test_gpu_qp.py -q -x > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
REPS=3 bash tools/r04_ab.sh r04q_ab "" "--opt asm_split=0"
EXTRA="--nvar 2048 --neq 0 --nineq 2048" STEPS=20 REPS=2 bash tools/r04_ab.sh r04q_ab2 "" "--opt asm_split=0"
