#!/bin/bash
set -u
O=gpurun_out/r04x; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $R/$O/prof -o qt -- python $R/tools/qp_solve.py > $R/$O/qp.json 2> $R/$O/qp.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/qp_timeline.py $DB $O/qp_timeline.txt
rm -rf $O/prof
