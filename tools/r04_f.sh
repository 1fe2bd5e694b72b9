#!/bin/bash
set -u
O=gpurun_out/r04f; mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_merit.py "tests/test_gpu_configs.py::test_oracle_lu_where_the_headline_runs" tests/test_gpu_pivoting.py tests/test_gpu_qp.py tests/test_gpu_ipm.py -q --durations=5 > $O/tests1.txt 2>&1
echo "tests1 rc=$?"; tail -12 $O/tests1.txt
REPS=2 bash tools/r04_ab.sh r04f_ab "" "--opt fuse_forward=0" "--opt reserve_cus=8" "--opt persist_rows=20480" "--opt persist_rows=16384" "--opt reserve_cus=24"
