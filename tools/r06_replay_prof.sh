#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=gpurun_out/r06p; mkdir -p $R/$O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof -o rp -- python $R/tools/rank_replay.py --nb 256 --worlds 8 --models sag --ranks 0 > $R/$O/replay.json 2> $R/$O/replay.err )
python tools/rocpd_stats.py $(find $O/prof -name "*.db" | head -1) $O/replay_kernel_stats.txt > /dev/null 2>&1; rm -rf $O/prof
head -30 $O/replay_kernel_stats.txt | cut -c1-150; tail -3 $O/replay.err
