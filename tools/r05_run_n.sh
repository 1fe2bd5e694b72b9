#!/bin/bash
set -u
O=gpurun_out/r05n; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYIPM_EXPERT=1
timeout 1200 python -m pytest tests/test_gpu_symmetric.py -x -q -k "bits or option" > $O/pytest_bits.log 2>&1; tail -3 $O/pytest_bits.log
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pivoting.py tests/test_gpu_ipm.py tests/test_gpu_qp.py -x -q > $O/pytest_par.log 2>&1; tail -3 $O/pytest_par.log
( timeout 300 python tools/ab_opts.py 2048 0 2048 30 "s_across=0" "" ) > $O/ab_cfg2.txt 2>&1; tail -2 $O/ab_cfg2.txt
( timeout 300 python tools/ab_opts.py 4096 1024 1536 20 "s_across=0" "" ) > $O/ab_8k.txt 2>&1; tail -2 $O/ab_8k.txt
( timeout 900 python tools/ab_opts.py 16384 4096 6144 4 "s_across=0" "" ) > $O/ab_metric.txt 2>&1; tail -2 $O/ab_metric.txt
( timeout 900 python tools/ab_opts.py 16384 8192 8192 2 "s_across=0" "" ) > $O/ab_cfg3.txt 2>&1; tail -2 $O/ab_cfg3.txt
