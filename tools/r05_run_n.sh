#!/bin/bash
set -u
O=gpurun_out/r05n; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1200 python -m pytest tests/test_gpu_tile_blocked.py -x -q > $O/pytest_tile.log 2>&1; tail -15 $O/pytest_tile.log
