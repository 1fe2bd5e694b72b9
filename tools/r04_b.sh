#!/bin/bash
set -u
O=gpurun_out/r04b; mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_merit.py "tests/test_gpu_configs.py::test_oracle_lu_where_the_headline_runs" -x -q --durations=5 > $O/new_tests.txt 2>&1
echo "new tests rc=$?" | tee -a $O/new_tests.txt; tail -15 $O/new_tests.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu.txt 2>&1
echo "gpu suite rc=$?" | tee -a $O/pytest_gpu.txt; tail -25 $O/pytest_gpu.txt
timeout 300 python tools/first_call.py > $O/first_call.json 2> $O/first_call.err; echo "first_call rc=$?"; cat $O/first_call.json
timeout 600 python tools/qp_solve.py > $O/qp_solve_full.json 2> $O/qp_solve_full.err
echo "qp_solve rc=$?"; tail -1 $O/qp_solve_full.json | head -c 2500; echo
