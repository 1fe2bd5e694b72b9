#!/bin/bash
# full GPU suite + headline benches (both settings of tile_blocked)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r03b}; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.txt
for b in 1 0; do
  timeout 300 python bench.py --nvar 2048 --neq 0 --nineq 2048 --steps 20 --warmup 3 --no-cpu-baseline --opt tile_blocked=$b > $O/cfg2_blocked$b.json 2>$O/cfg2_blocked$b.err; python -c "import json;d=json.load(open('$O/cfg2_blocked$b.json'));print('cfg2 blocked=$b ms/step',d['ms_per_step'],'berr',d['backward_error'],d['phases_ms_per_step'])"
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --opt tile_blocked=$b > $O/metric_blocked$b.json 2>$O/metric_blocked$b.err; python -c "import json;d=json.load(open('$O/metric_blocked$b.json'));print('metric blocked=$b ms/step',d['ms_per_step'],'berr',d['backward_error'],d['phases_ms_per_step'],d['roofline']['achieved'])"
done
