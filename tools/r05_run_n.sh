#!/bin/bash
set -u
O=gpurun_out/r05n; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYIPM_EXPERT=1
( timeout 1200 python tools/ab_opts.py 16384 4096 6144 4 "" "head_split_rows=12288" "head_split_rows=20480" "head_split_rows=12288,head_serial=2" "head_serial=2" ) > $O/ab_metric.txt 2>&1; tail -5 $O/ab_metric.txt
( timeout 900 python tools/ab_opts.py 16384 8192 8192 2 "" "head_split_rows=12288" "head_split_rows=20480" ) > $O/ab_cfg3.txt 2>&1; tail -3 $O/ab_cfg3.txt
