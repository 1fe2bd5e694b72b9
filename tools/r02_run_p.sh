#!/bin/bash
# config 2 timeline: which kernels sit on the chain and how long they and the gaps between them are
set -u
O=gpurun_out/r02p; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/$O/prof -o c2 -- python $R/bench.py --no-cpu-baseline --nvar 2048 --neq 0 --nineq 2048 --steps 3 --warmup 2 > $R/$O/c2.json 2> $R/$O/c2.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/c2_stats.txt | head -14
python tools/chain_timeline.py $DB 700 $O/c2_timeline.txt; head -5 $O/c2_timeline.txt
rm -rf $O/prof
