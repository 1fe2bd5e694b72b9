#!/bin/bash
# seeded sweeps beyond the suite's fixed cases with the round-5 defaults (eight-wave tile steps in the first group and the tail)
set -u
O=gpurun_out/r05f; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
( PYIPM_RAGGED_COUNT=300 PYIPM_RAGGED_SEED=505 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k ragged > $O/ragged.log 2>&1; tail -2 $O/ragged.log )
( PYIPM_RAGGED_COUNT=100 PYIPM_RAGGED_SEED=506 timeout 900 python -m pytest tests/test_gpu_condensed.py -q -x -k ragged > $O/ragged_condensed.log 2>&1; tail -2 $O/ragged_condensed.log )
( PYIPM_BACKEND_FUZZ=20,55 timeout 1200 python -m pytest tests/test_gpu_fuzz_backends.py -q -x > $O/backends.log 2>&1; tail -2 $O/backends.log )
( PYIPM_LP_FUZZ=15,57 timeout 900 python -m pytest tests/test_gpu_pivoting.py -q -x -k fuzz > $O/lp.log 2>&1; tail -2 $O/lp.log )
( PYIPM_DIST_FUZZ=8,58 timeout 900 python -m pytest tests/test_gpu_dist.py -q -x -k "fuzz or ragged or shapes" > $O/dist.log 2>&1; tail -2 $O/dist.log )
( PYIPM_BATCHED_FUZZ=30,59 timeout 600 python -m pytest tests/test_gpu_batched.py -q -x > $O/batched.log 2>&1; tail -2 $O/batched.log )
