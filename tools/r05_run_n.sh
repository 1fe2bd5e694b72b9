#!/bin/bash
set -u
O=gpurun_out/r05n; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYIPM_EXPERT=1
( timeout 300 python tools/tile_clock.py ) 2>&1 | grep -v amdgpu | grep -E "last tile|blocked path|k_tile_step block"
timeout 1200 python -m pytest tests/test_gpu_symmetric.py tests/test_gpu_tile_blocked.py -x -q -k "bits or option or eight" > $O/pytest_bits.log 2>&1; tail -2 $O/pytest_bits.log
( timeout 300 python tools/ab_opts.py 2048 0 2048 30 "" ) 2>&1 | tail -1
( timeout 900 python tools/ab_opts.py 16384 4096 6144 4 "" ) 2>&1 | tail -1
