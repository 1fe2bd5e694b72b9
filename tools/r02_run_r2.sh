#!/bin/bash
set -u
O=gpurun_out/r02r; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/$O/prof -o m -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 > $R/$O/m.json 2> $R/$O/m.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/chain_timeline.py $DB 300 $O/m_timeline.txt
grep -c k_bwd_step $O/m_timeline.txt
grep k_bwd_step $O/m_timeline.txt | awk '{print $2}' | tr '\n' ' ' | head -c 1500
rm -rf $O/prof
