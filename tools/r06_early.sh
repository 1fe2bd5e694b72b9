#!/bin/bash
# round 6: group 0's update panel by panel under its own chain (lookahead = 2) where the slack block follows it
set -u
O=gpurun_out/r06e; mkdir -p $O
( timeout 1800 python -m pytest tests/test_gpu_tile_blocked.py tests/test_gpu_symmetric.py tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -k "not lu_at" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log ); tail -5 $O/pytest.log
( timeout 600 python tools/ab_opts.py 2048 0 2048 30 "lookahead=1" "lookahead=2" > $O/ab_cfg2.txt 2>&1 ); grep -v amdgpu.ids $O/ab_cfg2.txt | tail -2
( timeout 600 python tools/ab_opts.py 3072 512 1024 30 "lookahead=1" "lookahead=2" > $O/ab_n5632.txt 2>&1 ); grep -v amdgpu.ids $O/ab_n5632.txt | tail -2
( timeout 900 python tools/ab_opts.py 16384 4096 6144 5 "lookahead=1" "lookahead=2" > $O/ab_n32768.txt 2>&1 ); grep -v amdgpu.ids $O/ab_n32768.txt | tail -2
( timeout 300 python tools/chain_stress.py 90 > $O/chain_stress.txt 2>&1 ); tail -1 $O/chain_stress.txt
