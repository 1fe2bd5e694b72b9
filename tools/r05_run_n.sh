#!/bin/bash
set -u
O=gpurun_out/r05n; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYIPM_EXPERT=1
( timeout 600 python tools/ab_opts.py 2048 0 2048 30 "" "tail_group=8" "tail_group=2" "tile_ny3=1" "early_head=1" "fuse_forward=0" "head_serial=1" "pending_left_rows=-1" ) > $O/ab_cfg2.txt 2>&1; tail -8 $O/ab_cfg2.txt
