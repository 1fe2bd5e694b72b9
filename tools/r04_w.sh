#!/bin/bash
set -u
O=gpurun_out/r04w; mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "suite rc=$?"; tail -3 $O/pytest.txt
timeout 300 python tools/bench_provider.py > $O/bench_provider.json 2>/dev/null; cat $O/bench_provider.json
timeout 600 python tools/qp_solve.py > $O/qp_solve_full.json 2> $O/qp.err
timeout 600 python tools/qp_solve.py --condensed > $O/qp_solve_condensed.json 2> $O/qpc.err
python -c "
import json
for f in ('qp_solve_full','qp_solve_condensed'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1])
    print(f, {k: d[k] for k in ('iterations','solve_seconds','newton_seconds','search_seconds','newton_seconds_per_factorisation')}, d['provider']['block_products_ms'])"
bash tools/pmc_hbm.sh $O/pmc_hbm > $O/pmc_hbm.log 2>&1; tail -3 $O/pmc_hbm.log
