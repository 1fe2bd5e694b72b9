#!/bin/bash
set -u
O=gpurun_out/r05d; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
( timeout 1500 python -X faulthandler -m pytest tests/test_gpu_dist.py -q --timeout 420 -o faulthandler_timeout=400 --deselect "tests/test_gpu_dist.py::test_ranks_sharing_one_gpu" > $O/pytest_dist.log 2>&1; echo "pytest rc=$?" >> $O/pytest_dist.log )
tail -40 $O/pytest_dist.log
( timeout 900 python -X faulthandler -m pytest tests/test_gpu_batched.py tests/test_gpu_pivoting.py tests/test_gpu_qp.py -q --timeout 240 > $O/pytest_bp.log 2>&1; echo "pytest rc=$?" >> $O/pytest_bp.log )
tail -60 $O/pytest_bp.log
