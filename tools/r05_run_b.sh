#!/bin/bash
set -u
O=gpurun_out/r05b; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
( timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_pivoting.py -m gpu -q --durations=3 > $O/pytest_b.log 2>&1; echo "pytest rc=$?" >> $O/pytest_b.log )
tail -8 $O/pytest_b.log
timeout 300 python tools/bench_batched.py > $O/bench_batched.txt 2>&1; cat $O/bench_batched.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_b -o b -- python $R/tools/bench_batched.py > /dev/null 2> $R/$O/rocprof_b.err )
python tools/rocpd_stats.py $(find $O/prof_b -name "*.db" | head -1) $O/batched_kernel_stats.txt > /dev/null 2>&1; rm -rf $O/prof_b
head -20 $O/batched_kernel_stats.txt
