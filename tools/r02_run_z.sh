#!/bin/bash
# round 2 evidence set: gpurun_out/r02z/ -> profiles/r02_z_*
set -u
O=gpurun_out/r02z; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -5 $O/pytest.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
timeout 900 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 > $R/$O/bench_under_rocprof.json 2> $R/$O/rocprof.err
cd $R
python tools/rocpd_stats.py $(find $O/prof -name "*.db" | head -1) $O/kernel_stats.txt > /dev/null 2>&1; head -12 $O/kernel_stats.txt
bash tools/pmc_update.sh $O/pmc > $O/pmc_update.log 2>&1
python tools/pmc_summary.py $O/pmc $O/pmc_update.json 2 > $O/pmc_summary.txt 2>&1; head -12 $O/pmc_summary.txt
bash tools/pmc_hbm.sh $O/pmc_hbm > $O/pmc_hbm.log 2>&1; tail -4 $O/pmc_hbm.log
bash tools/measure_configs.sh > $O/configs.txt 2>&1; tail -9 $O/configs.txt; mkdir -p $O/cfg; cp gpurun_out/cfg*.json gpurun_out/metric_forcedist.json $O/cfg/ 2>/dev/null
timeout 600 python tools/qp_solve.py > $O/qp_solve_full.json 2> $O/qp_solve_full.err
timeout 600 python tools/qp_solve.py --condensed > $O/qp_solve_condensed.json 2> $O/qp_solve_condensed.err
rm -rf $O/prof $O/pmc/*/*.db 2>/dev/null
du -sh $O
