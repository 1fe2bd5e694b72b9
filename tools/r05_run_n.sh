#!/bin/bash
set -u
O=gpurun_out/r05n; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYIPM_EXPERT=1
timeout 1200 python -m pytest tests/test_gpu_symmetric.py -x -q -k "bits or option" > $O/pytest_bits.log 2>&1; tail -3 $O/pytest_bits.log
( timeout 900 python tools/ab_opts.py 16384 4096 6144 4 "" "first_half=4" "first_half=2" "first_half=4,tile_waves=4" ) > $O/ab_metric.txt 2>&1; tail -4 $O/ab_metric.txt
( timeout 900 python tools/ab_opts.py 16384 8192 8192 2 "" "first_half=4" ) > $O/ab_cfg3.txt 2>&1; tail -2 $O/ab_cfg3.txt
