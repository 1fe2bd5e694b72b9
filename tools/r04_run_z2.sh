#!/bin/bash
# the counter / kernel-trace part of tools/r04_run_z.sh again (bench.py --no-clock: the traced process holds exactly warmup + steps steps),
# plus the tests added after the evidence set's pytest run
set -u
O=gpurun_out/r04z; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
kstats() {
  name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_$name -o $name -- python $R/bench.py --no-cpu-baseline --config4 off --no-clock "$@" > $R/$O/${name}_under_rocprof.json 2> $R/$O/${name}_rocprof.err < /dev/null )
  python tools/rocpd_stats.py "$(find $O/prof_$name -name '*.db' | head -1)" $O/${name}_kernel_stats.txt > /dev/null 2>&1 < /dev/null
  rm -rf $O/prof_$name
}
kstats bench --steps 4 --warmup 1; head -6 $O/bench_kernel_stats.txt
bash tools/pmc_update.sh $O/pmc > $O/pmc_update.log 2>&1 < /dev/null
python tools/pmc_summary.py $O/pmc $O/pmc_update.json 2 'k_update<256, true, 8>' > $O/pmc_summary.txt 2>&1; head -12 $O/pmc_summary.txt
python tools/pmc_summary.py $O/pmc $O/pmc_update_bn128.json 2 'k_update<128, true, 8>' > $O/pmc_summary_bn128.txt 2>&1
bash tools/pmc_hbm.sh $O/pmc_hbm > $O/pmc_hbm.log 2>&1 < /dev/null; tail -4 $O/pmc_hbm.log
kstats cfg2 --nvar 2048 --neq 0 --nineq 2048 --steps 10 --warmup 2
kstats cfg3 --nvar 16384 --neq 8192 --nineq 8192 --steps 2 --warmup 1
kstats cfg4 --nvar 65536 --neq 0 --nineq 32768 --steps 1 --warmup 1
rm -rf $O/pmc/*/*.db $O/pmc_hbm/*/*.db 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "ragged_shapes or block_matvec" 2>&1 | tail -3
