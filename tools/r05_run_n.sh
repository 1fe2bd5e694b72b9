#!/bin/bash
set -u
O=gpurun_out/r05n; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYIPM_EXPERT=1
( timeout 1200 python tools/ab_opts.py 16384 4096 6144 4 "" "bulk_waves=4" "bulk_bn_rows=12288,wide_persist_rows=20480" "bulk_bn_rows=12288,wide_persist_rows=20480,reserve_cus=8" "bulk_bn_rows=16384,wide_persist_rows=20480" "bulk_bn_rows=0,persist_rows=0,wide_persist_rows=12288" ) > $O/ab_metric.txt 2>&1; tail -6 $O/ab_metric.txt
