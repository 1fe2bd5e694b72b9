#!/bin/bash
# round 6: k_tile_chain against the launch-per-tile schedule -- bits (pytest) and in-process A/B timings
set -u
O=gpurun_out/r06a; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_tile_blocked.py -q -x -k "chain" > $O/pytest_chain.log 2>&1; echo "rc=$?" >> $O/pytest_chain.log )
tail -15 $O/pytest_chain.log
( timeout 600 python tools/ab_opts.py 2048 0 2048 30 "tile_chain=0" "tile_chain=1" "tile_chain=1,chain_whole=0" "tile_chain=1,chain_lds_kb=48" "tile_chain=1,chain_lds_kb=100" "tile_chain=1,tail_group=8" "tile_chain=1,tail_group=8,chain_lds_kb=100" > $O/ab_cfg2.txt 2>&1 ); cat $O/ab_cfg2.txt
( timeout 900 python tools/ab_opts.py 16384 4096 6144 5 "tile_chain=0" "tile_chain=1" "tile_chain=1,chain_lds_kb=100" "tile_chain=1,chain_lds_kb=48" > $O/ab_n32768.txt 2>&1 ); cat $O/ab_n32768.txt
( timeout 300 python tools/chain_clock.py 2048 0 2048 0 1 > $O/chain_clock_cfg2.txt 2>&1 ); cat $O/chain_clock_cfg2.txt
