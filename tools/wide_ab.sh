#!/bin/bash
# Per-panel schedule on one rank at nb = 1024 / 512 / 256 against the grouped single-GPU schedule, on ONE box (boxes differ
# by +-3 %), after the bitwise check of the wide-panel path (tools/wide_panel_check.py).  usage: gpurun -- bash tools/wide_ab.sh
timeout 600 python tools/wide_panel_check.py 2>&1 | grep "^n=" 
for o in "--nb 1024" "--nb 512" "--nb 256"; do
  timeout 600 python bench.py --steps 3 --warmup 1 --force-dist --no-cpu-baseline $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$o', round(d['ms_per_step'],2), d['phases_ms_per_step'], 'berr', d['backward_error'])"
done
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused', round(d['ms_per_step'],2), d['phases_ms_per_step'])"
timeout 1200 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
