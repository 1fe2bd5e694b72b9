#!/bin/bash
set -u
O=gpurun_out/r05e; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1200 python tools/rank_replay.py --nb 256 --worlds 8 > $O/replay_N32768_nb256_w8.json 2> $O/replay_N32768_nb256_w8.err; tail -5 $O/replay_N32768_nb256_w8.err
timeout 900 python tools/rank_replay.py --nb 256 --worlds 8 --slices 0 --models sag > $O/replay_N32768_nb256_w8_noslices.json 2> $O/replay_N32768_nb256_w8_noslices.err; tail -3 $O/replay_N32768_nb256_w8_noslices.err
( timeout 2400 python -m pytest tests -m gpu -q -x --timeout 600 --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -25 $O/pytest.log
