#!/bin/bash
set -u
O=gpurun_out/r04l; mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 600 python tools/opt_bits.py fused_head=1 > $O/opt_bits.txt 2>&1; grep -v amdgpu $O/opt_bits.txt
timeout 900 python -m pytest "tests/test_gpu_symmetric.py::test_random_schedule_options_give_the_same_bits" tests/test_gpu_qp.py tests/test_gpu_dist.py -q -x > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
REPS=3 bash tools/r04_ab.sh r04l_ab "" "--opt fused_head=1"
