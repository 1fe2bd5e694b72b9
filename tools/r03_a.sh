#!/bin/bash
# round 3, first GPU pass of the blocked tile inversion
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03a; mkdir -p $O
timeout 600 python tools/tile_blocked_check.py > $O/tile_check.txt 2>&1; echo "tile_check rc=$?"
tail -20 $O/tile_check.txt
for b in 1 0; do
  sed "s/core.set_option(\"lookahead\", 0)/core.set_option(\"lookahead\", 0); core.set_option(\"tile_blocked\", $b)/" tools/tile_clock.py > /tmp/tc.py
  cp /tmp/tc.py tools/_tc_tmp.py; timeout 300 python tools/_tc_tmp.py 2>&1 | tail -1; rm -f tools/_tc_tmp.py
done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_symmetric.py tests/test_gpu_pivoting.py tests/test_gpu_condensed.py tests/test_gpu_batched.py tests/test_gpu_ipm.py -x -q > $O/pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_subset.txt
for b in 1 0; do
  timeout 300 python bench.py --nvar 2048 --neq 0 --nineq 2048 --steps 20 --warmup 3 --no-cpu-baseline --opt tile_blocked=$b > $O/cfg2_blocked$b.json 2>$O/cfg2_blocked$b.err; python -c "import json;d=json.load(open('$O/cfg2_blocked$b.json'));print('cfg2 blocked=$b ms/step',d['ms_per_step'],'berr',d['backward_error'],d['phases_ms_per_step'])"
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --opt tile_blocked=$b > $O/metric_blocked$b.json 2>$O/metric_blocked$b.err; python -c "import json;d=json.load(open('$O/metric_blocked$b.json'));print('metric blocked=$b ms/step',d['ms_per_step'],'berr',d['backward_error'],d['phases_ms_per_step'],d['roofline']['achieved'])"
done
