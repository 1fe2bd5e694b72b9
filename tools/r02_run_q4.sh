#!/bin/bash
set -u
O=gpurun_out/r02q; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_symmetric.py tests/test_gpu_parity.py tests/test_gpu_condensed.py tests/test_gpu_pivoting.py tests/test_gpu_ipm.py tests/test_gpu_qp.py -m gpu -q > $O/pytest_a.log 2>&1; tail -5 $O/pytest_a.log
bash tools/r02_run_q3.sh
