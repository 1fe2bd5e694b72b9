#!/bin/bash
# 8-wave tile step: parity (tile families, bitwise-neutrality sweep), the tile clock, config 2 / headline A/B, replay
set -u
O=gpurun_out/r05m; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYIPM_EXPERT=1
timeout 900 python -m pytest tests/test_gpu_tile_blocked.py -x -q > $O/pytest_tile.log 2>&1; tail -3 $O/pytest_tile.log
timeout 1200 python -m pytest tests/test_gpu_symmetric.py -x -q -k "bits or option" > $O/pytest_bits.log 2>&1; tail -3 $O/pytest_bits.log
( timeout 300 python tools/tile_clock.py ) > $O/tile_clock8.txt 2>&1; grep -v amdgpu.ids $O/tile_clock8.txt | grep -v "wave [123]" | head -24
( timeout 300 python tools/ab_opts.py 2048 0 2048 30 "tile_waves=4" "tile_waves=8" ) > $O/ab_cfg2.txt 2>&1; tail -2 $O/ab_cfg2.txt
( timeout 900 python tools/ab_opts.py 16384 4096 6144 4 "tile_waves=4" "tile_waves=8" ) > $O/ab_metric.txt 2>&1; tail -2 $O/ab_metric.txt
for w in 4 8; do timeout 600 python tools/rank_replay.py --nb 256 --worlds 8 --models sag --opt tile_waves=$w > $O/replay_w$w.json 2> $O/replay_w$w.err; tail -1 $O/replay_w$w.err; done
