#!/bin/bash
set -u
O=gpurun_out/r05n; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYIPM_EXPERT=1
python tools/group_trace.py 2048 0 2048 "early_head=1" 2>&1 | grep -v amdgpu | tail -36 | grep -E "first group|g0|g1|begin g[45]|end g[45]|factorisation"
( timeout 300 python tools/ab_opts.py 2048 0 2048 30 "" "early_head=1" "early_head=1,early_first=0" ) > $O/ab_cfg2.txt 2>&1; tail -3 $O/ab_cfg2.txt
( timeout 300 python tools/ab_opts.py 4096 1024 1536 20 "" "early_head=1" "early_head=1,early_first=0" ) > $O/ab_8k.txt 2>&1; tail -3 $O/ab_8k.txt
( timeout 900 python tools/ab_opts.py 16384 4096 6144 4 "" "early_head=1" "early_head=1,early_first=0" ) > $O/ab_metric.txt 2>&1; tail -3 $O/ab_metric.txt
