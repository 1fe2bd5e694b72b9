#!/bin/bash
# gpurun_out/r06z/* (tools/r06_run_z.sh) -> profiles/r06_z_*
S=gpurun_out/r06z P=profiles/r06_z
cp $S/bench.json ${P}_bench.json; cp $S/bench_default.json ${P}_bench_default.json; cp $S/bench_under_rocprof.json ${P}_bench_under_rocprof.json
cp $S/bench_kernel_stats.txt ${P}_kernel_stats.txt; cp $S/pmc_update.json ${P}_pmc_update.json; cp $S/pmc_update_bn128.json ${P}_pmc_update_bn128.json; cp $S/pmc_hbm/pmc_hbm.json ${P}_pmc_hbm_kernels.json
for c in 2 3 4; do cp $S/cfg${c}_kernel_stats.txt ${P}_cfg${c}_kernel_stats.txt; done
for f in $S/cfg/*.json; do cp $f ${P}_$(basename $f); done
for f in bench_batched.txt bench_tile.txt update_cycles.txt chain_clock_cfg2.txt group_trace_cfg2.txt ab_chain_cfg2.txt ab_chain_n32768.txt; do [ -f $S/$f ] && grep -v "amdgpu.ids" $S/$f > ${P}_$f; done
for f in qp_solve_condensed qp_solve_full replay_N32768 replay_N131072 first_call bench_provider; do [ -f $S/$f.json ] && grep '^{' $S/$f.json | tail -1 > ${P}_$f.json; done
cp $S/qp_kernel_stats.txt ${P}_qp_search_kernel_stats.txt
grep '^{' $S/bench_lbfgs.txt | tail -1 > ${P}_lbfgs.json
grep pyipm $S/first_call.err > ${P}_first_call_setup_trace.txt
tail -5 $S/pytest.log > ${P}_pytest_gpu_tail.txt
for f in replay_N32768_link50 replay_N32768_link150 replay_N32768_tile_chain0 replay_N65536_bitwise replay_N131072_link75 replay_N131072_link50 replay_N131072_link150; do [ -f $S/$f.json ] && grep '^{' $S/$f.json | tail -1 > ${P}_$f.json; done
[ -f $S/batched_kernel_stats.txt ] && cp $S/batched_kernel_stats.txt ${P}_batched_kernel_stats.txt
