#!/bin/bash
# PMC passes for the trailing-update kernel (separate runs per counter group; gfx950 slot limits:
# SQ 8, TCC 4 (FETCH_SIZE costs 3, WRITE_SIZE 2), GRBM 2).  Usage: tools/pmc_update.sh <outdir> [bench args]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --config4 off --configs off --no-clock $*"
run() { name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "k_update" --output-format csv -d $ROOT/$OUT/$name -o $name -- python $ROOT/bench.py $ARGS > $ROOT/$OUT/$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls -R $ROOT/$OUT | head -40
