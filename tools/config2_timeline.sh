#!/bin/bash
# Kernel trace of config 2 (chain-bound) -> gpurun_out/r02q/c2_timeline.txt: every launch of the last steps in start order
# with its duration and the gap to the previous launch of the same queue (tools/chain_timeline.py).  Extra args go to bench.py.
set -u
O=gpurun_out/${OUTDIR:-r03q}; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/$O/prof -o c2 -- python $R/bench.py --no-cpu-baseline --nvar 2048 --neq 0 --nineq 2048 --steps 3 --warmup 2 $@ > $R/$O/c2.json 2> $R/$O/c2.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/chain_timeline.py $DB 500 $O/c2_timeline.txt
rm -rf $O/prof
