#!/bin/bash
# round 3 evidence set: gpurun_out/r03z/ -> profiles/r03_z_*   (run on the GPU box: gpurun -- bash tools/r03_run_z.sh)
set -u
O=gpurun_out/r03z; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
( timeout 3000 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -5 $O/pytest.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json
timeout 900 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
# kernel trace + stats of the default command
kstats() {   # name, bench args...
  name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_$name -o $name -- python $R/bench.py --no-cpu-baseline "$@" > $R/$O/${name}_under_rocprof.json 2> $R/$O/${name}_rocprof.err )
  python tools/rocpd_stats.py $(find $O/prof_$name -name "*.db" | head -1) $O/${name}_kernel_stats.txt > /dev/null 2>&1
  rm -rf $O/prof_$name
}
kstats bench --steps 4 --warmup 1; head -12 $O/bench_kernel_stats.txt
bash tools/pmc_update.sh $O/pmc > $O/pmc_update.log 2>&1
python tools/pmc_summary.py $O/pmc $O/pmc_update.json 2 'k_update<256, true, 8>' > $O/pmc_summary.txt 2>&1; head -12 $O/pmc_summary.txt
python tools/pmc_summary.py $O/pmc $O/pmc_update_bn128.json 2 'k_update<128, true, 8>' > $O/pmc_summary_bn128.txt 2>&1
bash tools/pmc_hbm.sh $O/pmc_hbm > $O/pmc_hbm.log 2>&1; tail -4 $O/pmc_hbm.log
# the other BASELINE configurations: bench line, kernel stats, one counter pass each (FETCH / WRITE / MFMA busy)
bash tools/measure_configs.sh > $O/configs.txt 2>&1; tail -9 $O/configs.txt; mkdir -p $O/cfg; cp gpurun_out/cfg*.json gpurun_out/metric_*.json $O/cfg/ 2>/dev/null
cfgpmc() {   # name, bench args...
  name=$1; shift
  pass() { p=$1; ctrs=$2; shift 2
    ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/$O/pmc_${name}/$p -o $p -- python $R/bench.py --no-cpu-baseline "$@" > $R/$O/pmc_${name}_$p.log 2>&1 )
  }
  pass tcc1 "FETCH_SIZE GRBM_GUI_ACTIVE" "$@"
  pass tcc2 "WRITE_SIZE" "$@"
  pass sq1 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" "$@"
  python tools/pmc_config_summary.py $O/pmc_${name} $O/${name}_pmc.json > $O/${name}_pmc.txt 2>&1; head -8 $O/${name}_pmc.txt
  rm -rf $O/pmc_${name}
}
kstats cfg2 --nvar 2048 --neq 0 --nineq 2048 --steps 10 --warmup 2
cfgpmc cfg2 --nvar 2048 --neq 0 --nineq 2048 --steps 2 --warmup 1
kstats cfg3 --nvar 16384 --neq 8192 --nineq 8192 --steps 2 --warmup 1
cfgpmc cfg3 --nvar 16384 --neq 8192 --nineq 8192 --steps 1 --warmup 1
kstats cfg4 --nvar 65536 --neq 0 --nineq 32768 --steps 1 --warmup 1
cfgpmc cfg4 --nvar 65536 --neq 0 --nineq 32768 --steps 1 --warmup 0
timeout 300 python tools/tile_clock.py 1 > $O/tile_clock_blocked.txt 2>&1; timeout 300 python tools/tile_clock.py 0 | head -1 > $O/tile_clock_sweeps.txt 2>&1
( timeout 300 python tools/bench_tile.py 1; timeout 300 python tools/bench_tile.py 0 ) > $O/bench_tile.txt 2>&1
timeout 300 python tools/tile_blocked_check.py > $O/tile_blocked_check.txt 2>&1
( timeout 300 python tools/contention_probe.py; timeout 300 python tools/contention_probe.py bulk_bn=128 ) > $O/contention_probe.txt 2>&1
( timeout 300 tools/ubench/contention 128 16384; timeout 300 tools/ubench/contention 256 16384 ) > $O/contention_ubench.txt 2>&1
( timeout 300 python tools/sweep_clock.py; timeout 300 python tools/sweep_clock.py 16384 4096 6144 ) > $O/sweep_clock.txt 2>&1
timeout 600 python tools/sweep_check.py > $O/sweep_check.txt 2>&1
timeout 600 python tools/fwd_check.py big > $O/fwd_check.txt 2>&1
timeout 120 tools/ubench/pingpong > $O/pingpong.txt 2>&1
( timeout 300 tools/ubench/rounds 40 64; timeout 300 tools/ubench/rounds 58 66 256; timeout 300 tools/ubench/rounds 126 128 256; timeout 300 tools/ubench/rounds 126 128 128 ) > $O/rounds.txt 2>&1
timeout 600 python tools/qp_solve.py > $O/qp_solve_full.json 2> $O/qp_solve_full.err
timeout 600 python tools/qp_solve.py --condensed > $O/qp_solve_condensed.json 2> $O/qp_solve_condensed.err
timeout 600 python tools/bench_batched.py > $O/bench_batched.txt 2>&1
timeout 900 python tools/bench_lbfgs.py > $O/bench_lbfgs.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --steps 4 --warmup 2 --opt bulk_bn=128 > $O/bench_bulk_bn128.json 2> /dev/null
timeout 600 python bench.py --no-cpu-baseline --steps 4 --warmup 2 --opt tile_blocked=0 > $O/bench_tile_sweeps.json 2> /dev/null
rm -rf $O/pmc/*/*.db $O/pmc_hbm/*/*.db 2>/dev/null
du -sh $O
