#!/bin/bash
set -u
O=gpurun_out/r05f; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python tools/opt_bits_small.py > $O/opt_bits_small.txt 2>&1; cat $O/opt_bits_small.txt | grep -v amdgpu.ids
( timeout 900 python -m pytest tests/test_gpu_dist.py -q -x --timeout 420 -k "exchange_forms or selftest or sharing_one_gpu_ragged or drivers_agree" > $O/pytest_dist.log 2>&1; echo "pytest rc=$?" >> $O/pytest_dist.log ); tail -5 $O/pytest_dist.log
timeout 1200 python tools/rank_replay.py --nb 256 --worlds 8 --models sag > $O/replay_N32768_nb256_w8.json 2> $O/replay_N32768_nb256_w8.err; tail -3 $O/replay_N32768_nb256_w8.err
timeout 600 python tools/ab_opts.py 2048 0 2048 30 "" "fuse_forward=0" "tail_group=8" "tail_group=2" "pending_left_rows=-1" "head32_rows=1048576" "tail_group=8,fuse_forward=0" > $O/ab_cfg2.txt 2>&1; grep -v amdgpu $O/ab_cfg2.txt
