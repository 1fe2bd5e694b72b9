#!/bin/bash
# round 6 evidence set: gpurun_out/r06z/ -> profiles/r06_z_*   (on the GPU box: gpurun -- "WITH_REPLAY=1 bash tools/r06_run_z.sh"; then tools/r06_copy_z.sh here)
set -u
O=gpurun_out/r06z; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
( timeout 3000 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -5 $O/pytest.log
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
timeout 900 python bench.py --no-cpu-baseline --config4 off --configs off --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
# kernel trace + stats of the headline workload (the default command without its config-4 leg and CPU leg)
kstats() {   # name, bench args...
  name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_$name -o $name -- python $R/bench.py --no-cpu-baseline --config4 off --configs off --no-clock "$@" > $R/$O/${name}_under_rocprof.json 2> $R/$O/${name}_rocprof.err )
  python tools/rocpd_stats.py $(find $O/prof_$name -name "*.db" | head -1) $O/${name}_kernel_stats.txt > /dev/null 2>&1
  rm -rf $O/prof_$name
}
kstats bench --steps 4 --warmup 1; head -12 $O/bench_kernel_stats.txt
bash tools/pmc_update.sh $O/pmc > $O/pmc_update.log 2>&1
python tools/pmc_summary.py $O/pmc $O/pmc_update.json 2 'k_update<256, true, 8>' > $O/pmc_summary.txt 2>&1; head -12 $O/pmc_summary.txt
python tools/pmc_summary.py $O/pmc $O/pmc_update_bn128.json 2 'k_update<128, true, 8>' > $O/pmc_summary_bn128.txt 2>&1
bash tools/pmc_hbm.sh $O/pmc_hbm > $O/pmc_hbm.log 2>&1; tail -4 $O/pmc_hbm.log
bash tools/measure_configs.sh > $O/configs.txt 2>&1; tail -11 $O/configs.txt; mkdir -p $O/cfg; cp gpurun_out/cfg*.json gpurun_out/metric_*.json $O/cfg/ 2>/dev/null
kstats cfg2 --nvar 2048 --neq 0 --nineq 2048 --steps 10 --warmup 2
kstats cfg3 --nvar 16384 --neq 8192 --nineq 8192 --steps 2 --warmup 1
kstats cfg4 --nvar 65536 --neq 0 --nineq 32768 --steps 1 --warmup 1
timeout 600 python tools/qp_solve.py > $O/qp_solve_full.json 2> $O/qp_solve_full.err
timeout 600 python tools/qp_solve.py --condensed > $O/qp_solve_condensed.json 2> $O/qp_solve_condensed.err
# the search phase of the device QP loop under a kernel trace: no at::native reduction left in it
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_qp -o qp -- python $R/tools/qp_solve.py --nvar 4096 --neq 1024 --nineq 1536 > $R/$O/qp_small_under_rocprof.json 2> $R/$O/qp_rocprof.err )
python tools/rocpd_stats.py $(find $O/prof_qp -name "*.db" | head -1) $O/qp_kernel_stats.txt > /dev/null 2>&1; rm -rf $O/prof_qp
if [ -n "${WITH_REPLAY:-}" ]; then bash tools/r06_run_z_replay.sh; fi
PYIPM_SETUP_TRACE=1 timeout 300 python tools/first_call.py > $O/first_call.json 2> $O/first_call.err
timeout 300 python tools/bench_provider.py > $O/bench_provider.json 2> /dev/null
timeout 600 python tools/bench_batched.py > $O/bench_batched.txt 2>&1
timeout 900 python tools/bench_lbfgs.py > $O/bench_lbfgs.txt 2>&1
( timeout 300 python tools/bench_tile.py 1 ) > $O/bench_tile.txt 2>&1
( timeout 300 python tools/chain_clock.py 2048 0 2048 0 1 ) > $O/chain_clock_cfg2.txt 2>&1
( timeout 300 python tools/group_trace.py 2048 0 2048 2>&1 | tail -45 ) > $O/group_trace_cfg2.txt 2>&1
( timeout 600 python tools/ab_opts.py 2048 0 2048 30 "tile_chain=0" "" "chain_lds_kb=0" "chain_whole=0" "lookahead=1" "lookahead=1,bulk_bn=128" ) > $O/ab_chain_cfg2.txt 2>&1
( timeout 900 python tools/ab_opts.py 16384 4096 6144 5 "tile_chain=0" "" "tile_chain=2" ) > $O/ab_chain_n32768.txt 2>&1
( timeout 300 python tools/update_cycles.py; echo; echo '--- one wide launch (K = 1024) repeated in isolation:'; timeout 300 python tools/timeline_update.py 1024 | head -8 ) > $O/update_cycles.txt 2>&1 < /dev/null
rm -rf $O/pmc/*/*.db $O/pmc_hbm/*/*.db 2>/dev/null
du -sh $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_b -o b -- python $R/tools/bench_batched.py > /dev/null 2> $R/$O/rocprof_b.err )
python tools/rocpd_stats.py $(find $O/prof_b -name "*.db" | head -1) $O/batched_kernel_stats.txt > /dev/null 2>&1; rm -rf $O/prof_b
du -sh $O
