#!/bin/bash
# Kernel trace of the device QP loop -> per-iterate summary (tools/qp_timeline.py).  usage (GPU box): bash tools/qp_timeline.sh [outdir]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=${1:-gpurun_out/qptl}; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $R/$O/prof -o qt -- python $R/tools/qp_solve.py > $R/$O/qp.json 2> $R/$O/qp.err < /dev/null
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/qp_timeline.py "$DB" $O/qp_timeline.txt < /dev/null
rm -rf $O/prof
