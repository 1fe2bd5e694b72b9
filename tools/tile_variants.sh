#!/bin/bash
# Timing experiments on the pivot loop of k_tile_invert: extra libraries with parts of the loop removed (WRONG results),
# each timed with tools/tile_clock.py.  Build here (hipcc cross-compiles), run on the GPU box.
# usage: tools/tile_variants.sh build | run
set -u
ROOT=$(cd $(dirname $0)/.. && pwd)
V="0 1 2 4 8 3 7 15"
if [ "${1:-run}" = build ]; then
  mkdir -p $ROOT/gpurun_in
  for b in $V; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DPYIPM_TILE_EXPERIMENT=$b \
      -o $ROOT/gpurun_in/libpyipm_tx$b.so $ROOT/pyipm_amd/csrc/pyipm_newton.hip 2>/dev/null &
  done; wait; ls -la $ROOT/gpurun_in
else
  for b in $V; do
    echo -n "experiment bits=$b: "; PYIPM_NEWTON_LIB=$ROOT/gpurun_in/libpyipm_tx$b.so python $ROOT/tools/tile_clock.py 2>&1 | grep "last tile"
  done
fi
