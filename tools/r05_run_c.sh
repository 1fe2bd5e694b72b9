#!/bin/bash
set -u
O=gpurun_out/r05c; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
( timeout 300 python -X faulthandler -m pytest tests/test_gpu_pivoting.py -q -x -k "unconverged" -o faulthandler_timeout=100 > $O/pytest_hang.log 2>&1; echo "pytest rc=$?" >> $O/pytest_hang.log )
tail -60 $O/pytest_hang.log
timeout 200 python tools/bench_batched.py > $O/bench_batched.txt 2>&1; cat $O/bench_batched.txt
( timeout 1500 python -X faulthandler -m pytest tests/test_gpu_dist.py -q -x --timeout 420 -o faulthandler_timeout=400 > $O/pytest_dist.log 2>&1; echo "pytest rc=$?" >> $O/pytest_dist.log )
tail -40 $O/pytest_dist.log
