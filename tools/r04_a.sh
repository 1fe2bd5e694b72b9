#!/bin/bash
# round 4, first GPU call: the new tests, the whole GPU suite, the default bench line (with the config-4 leg), a device QP solve
set -u
O=gpurun_out/r04a; mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_merit.py tests/test_gpu_qp.py "tests/test_gpu_symmetric.py::test_wide_bulk_tiles_stop_at_the_storage_edge" \
    "tests/test_gpu_pivoting.py" "tests/test_gpu_configs.py::test_oracle_lu_where_the_headline_runs" -x -q --durations=8 > $O/new_tests.txt 2>&1
echo "new tests rc=$?" | tee -a $O/new_tests.txt
tail -30 $O/new_tests.txt
timeout 1200 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest_gpu.txt 2>&1
echo "gpu suite rc=$?" | tee -a $O/pytest_gpu.txt
tail -25 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"; head -c 1500 $O/bench_default.json; echo
timeout 600 python tools/qp_solve.py > $O/qp_solve_full.json 2> $O/qp_solve_full.err
echo "qp_solve rc=$?"; tail -1 $O/qp_solve_full.json | head -c 2500; echo
timeout 600 python tools/qp_solve.py --condensed > $O/qp_solve_condensed.json 2> $O/qp_solve_condensed.err
echo "qp_solve condensed rc=$?"; tail -1 $O/qp_solve_condensed.json | head -c 1200; echo
