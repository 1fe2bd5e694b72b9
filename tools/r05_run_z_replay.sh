#!/bin/bash
# the rank replays of the round-5 evidence set (gpurun_out/r05z/replay_*), apart from tools/r05_run_z.sh: they take the longest
set -u
O=gpurun_out/r05z; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1500 python tools/rank_replay.py --nb 256 > $O/replay_N32768.json 2> $O/replay_N32768.err; tail -6 $O/replay_N32768.err
timeout 900 python tools/rank_replay.py --nb 256 --worlds 8 --models sag --slices 0 > $O/replay_N32768_one_message.json 2> /dev/null
timeout 900 python tools/rank_replay.py --nb 256 --worlds 8 --models sag --serialize 1 > $O/replay_N32768_one_communicator.json 2> /dev/null
timeout 1200 python tools/rank_replay.py --nb 1024 --worlds 8 --models sag > $O/replay_N32768_nb1024.json 2> /dev/null
timeout 3000 python tools/rank_replay.py --nvar 65536 --neq 0 --nineq 32768 --host-record --steps 1 --worlds 4,8 --models sag > $O/replay_N131072.json 2> $O/replay_N131072.err; tail -6 $O/replay_N131072.err
