#!/bin/bash
set -u
O=gpurun_out/r04u; mkdir -p $O
cd "$(dirname "$0")/.."
for i in 1 2 3; do timeout 900 python -m pytest "tests/test_gpu_symmetric.py::test_random_schedule_options_give_the_same_bits" -q -x > $O/rand_$i.txt 2>&1; echo "random options run $i rc=$?"; done
bash tools/r04_t.sh
