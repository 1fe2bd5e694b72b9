#!/bin/bash
set -u
O=gpurun_out/r05n; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYIPM_EXPERT=1
( timeout 1200 python tools/ab_opts.py 16384 4096 6144 4 "" "reserve_cus=32" "reserve_cus=48" "reserve_cus=64" "reserve_cus=96" "reserve_cus=64,head_serial=2" "reserve_cus=48,persist_rows=10240" ) > $O/ab_metric.txt 2>&1; tail -7 $O/ab_metric.txt
