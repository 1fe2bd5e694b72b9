#!/bin/bash
# Kernel trace of the headline workload -> $OUT/metric_timeline.txt (every launch of the last step; tools/chain_timeline.py)
set -u
O=gpurun_out/${OUTDIR:-r03d}; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/$O/prof -o mt -- python $R/bench.py --no-cpu-baseline --no-clock --config4 off --steps 2 --warmup 1 $@ > $R/$O/mt.json 2> $R/$O/mt.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/chain_timeline.py $DB 2600 $O/metric_timeline.txt
rm -rf $O/prof
