#!/bin/bash
# gpurun_out/r03z/* (tools/r03_run_z.sh) -> profiles/r03_z_*
S=gpurun_out/r03z P=profiles/r03_z
cp $S/bench.json ${P}_bench.json; cp $S/bench_default.json ${P}_bench_default.json; cp $S/bench_under_rocprof.json ${P}_bench_under_rocprof.json
cp $S/bench_kernel_stats.txt ${P}_kernel_stats.txt; cp $S/pmc_update.json ${P}_pmc_update.json; cp $S/pmc_update_bn128.json ${P}_pmc_update_bn128.json; cp $S/pmc_hbm/pmc_hbm.json ${P}_pmc_hbm_kernels.json
for c in 2 3 4; do cp $S/cfg${c}_kernel_stats.txt ${P}_cfg${c}_kernel_stats.txt; cp $S/cfg${c}_pmc.json ${P}_cfg${c}_pmc.json; done
for f in $S/cfg/*.json; do cp $f ${P}_$(basename $f); done
for f in bench_batched.txt bench_tile.txt contention_probe.txt contention_ubench.txt tile_blocked_check.txt tile_clock_blocked.txt tile_clock_sweeps.txt sweep_clock.txt sweep_check.txt fwd_check.txt pingpong.txt rounds.txt; do
  [ -f $S/$f ] && grep -v "amdgpu.ids" $S/$f > ${P}_$f; done
for f in bench_bulk_bn128.json bench_tile_sweeps.json; do cp $S/$f ${P}_$f; done
for f in qp_solve_condensed qp_solve_full; do grep '^{' $S/$f.json > ${P}_$f.json; done
grep '^{' $S/bench_lbfgs.txt | tail -1 > ${P}_lbfgs.json
tail -5 $S/pytest.log > ${P}_pytest_gpu_tail.txt
