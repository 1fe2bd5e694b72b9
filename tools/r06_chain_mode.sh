#!/bin/bash
# round 6: the chain's step after the first form (inverse kept in shared memory, early look, deferred word) -- bits, then timings
set -u
O=gpurun_out/r06m; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_tile_blocked.py -q -x > $O/pytest_tile.log 2>&1; echo "rc=$?" >> $O/pytest_tile.log ); tail -3 $O/pytest_tile.log
( timeout 600 python tools/ab_opts.py 2048 0 2048 30 "tile_chain=1" "tile_chain=1,chain_whole=0" "tile_chain=0" > $O/ab_cfg2.txt 2>&1 ); grep -v amdgpu.ids $O/ab_cfg2.txt | tail -4
( timeout 300 python tools/chain_clock.py 2048 0 2048 0 1 > $O/chain_clock_cfg2.txt 2>&1 ); grep -v amdgpu.ids $O/chain_clock_cfg2.txt | sed -n 40,62p | cut -c1-100
( timeout 900 python tools/ab_opts.py 16384 4096 6144 5 "tile_chain=1" "tile_chain=0" > $O/ab_n32768.txt 2>&1 ); grep -v amdgpu.ids $O/ab_n32768.txt | tail -3
( timeout 600 python tools/chain_stress.py 120 > $O/chain_stress.txt 2>&1 ); tail -3 $O/chain_stress.txt
