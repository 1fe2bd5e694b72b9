#!/bin/bash
set -u
O=gpurun_out/r04e; mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_merit.py "tests/test_gpu_symmetric.py::test_no_kernel_reads_memory_nobody_wrote" "tests/test_gpu_symmetric.py::test_wide_bulk_tiles_stop_at_the_storage_edge" "tests/test_gpu_dist.py::test_wide_bulk_tiles_across_ranks" "tests/test_gpu_dist.py::test_ranks_sharing_one_gpu" -q --durations=5 > $O/tests1.txt 2>&1
echo "tests1 rc=$?"; tail -12 $O/tests1.txt
timeout 600 python tools/opt_bits.py ingroup_waves=8 ingroup_waves=4 bulk_split=1 bulk_split=1,ingroup_waves=8 > $O/opt_bits.txt 2>&1; cat $O/opt_bits.txt | grep -v amdgpu
REPS=2 bash tools/r04_ab.sh r04e_ab "" "--opt ingroup_waves=8" "--opt ingroup_waves=4" "--opt bulk_split=1" "--opt bulk_split=1 --opt ingroup_waves=8"
