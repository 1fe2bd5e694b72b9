#!/bin/bash
# 8-wave tile step: parity (tile families, bitwise-neutrality sweep), the tile clock, config 2 / headline A/B
set -u
O=gpurun_out/r05m; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYIPM_EXPERT=1
timeout 900 python -m pytest tests/test_gpu_tile_blocked.py -x -q > $O/pytest_tile.log 2>&1; tail -3 $O/pytest_tile.log
timeout 1200 python -m pytest tests/test_gpu_symmetric.py -x -q -k "bits or option" > $O/pytest_bits.log 2>&1; tail -3 $O/pytest_bits.log
( timeout 300 python tools/tile_clock.py ) > $O/tile_clock8.txt 2>&1; grep -v amdgpu.ids $O/tile_clock8.txt | grep -v "wave [123]" | head -24
( timeout 300 python tools/ab_opts.py 2048 0 2048 30 "tile_waves=4" "tile_waves=8" "tile_waves=8,tile_free_cus=256" "tile_waves=8,tile_free_cus=64" "tile_waves=8,tile_ny3=1" "tile_waves=8,tile_ny3=1,tile_free_cus=64" ) > $O/ab_cfg2.txt 2>&1; tail -6 $O/ab_cfg2.txt
( timeout 900 python tools/ab_opts.py 16384 4096 6144 4 "tile_waves=4" "tile_waves=8" "tile_waves=8,tile_free_cus=64" "tile_waves=8,tile_ny3=1" "tile_waves=8,tile8_rows=6144" "tile_waves=8,tile8_rows=0" "tile_waves=8,tile8_rows=20480" ) > $O/ab_metric.txt 2>&1; tail -7 $O/ab_metric.txt
