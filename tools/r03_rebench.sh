#!/bin/bash
# the three bench lines + kernel statistics of the evidence set again (after the PMC summaries they quote have been copied
# to profiles/): gpurun_out/r03z/bench*.json
set -u
O=gpurun_out/r03z; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.json
timeout 900 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_rocprof.err )
python tools/rocpd_stats.py $(find $O/prof_bench -name "*.db" | head -1) $O/bench_kernel_stats.txt > /dev/null 2>&1
rm -rf $O/prof_bench
head -6 $O/bench_kernel_stats.txt | cut -c1-150
