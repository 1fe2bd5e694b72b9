#!/bin/bash
# A/B of two builds of the library on config 2 (N = 6144, chain-bound) and the headline, interleaved on one box.
# usage: tools/ab_cfg2.sh LIB_A.so LIB_B.so   (PYIPM_NEWTON_LIB selects the build)
set -u
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for lib in "$@"; do
    c2=$(PYIPM_NEWTON_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --config4 off --nvar 2048 --neq 0 --nineq 2048 --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])")
    hd=$(PYIPM_NEWTON_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --config4 off --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f (panel %.2f)' % (d['ms_per_step'], d['phases_ms_per_step']['panel(tile+scale+in-panel)']))")
    echo "$(basename $lib): config 2 $c2 ms, headline $hd"
  done
done
