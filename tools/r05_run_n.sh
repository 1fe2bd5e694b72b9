#!/bin/bash
set -u
O=gpurun_out/r05n; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYIPM_EXPERT=1
( timeout 900 python tools/ab_opts.py 16384 4096 6144 4 "" "bulk_bn_rows=16384" "bulk_bn_rows=12288" "bulk_bn_rows=16384,persist_rows=10240" "tile_waves=4" ) > $O/ab_metric.txt 2>&1; tail -5 $O/ab_metric.txt
