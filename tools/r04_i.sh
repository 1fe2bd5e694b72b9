#!/bin/bash
set -u
O=gpurun_out/r04i; mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
PYIPM_SETUP_TRACE=1 timeout 300 python tools/first_call.py > $O/first_call.json 2> $O/first_call.err; grep pyipm $O/first_call.err
python -c "
import json; d=json.load(open('$O/first_call.json'))
for c in d['calls']: print({k: round(v,2) for k,v in c.items() if k.endswith('_ms')})"
timeout 900 python -m pytest tests/test_gpu_dist.py -q -x > $O/tests_dist.txt 2>&1; echo "dist tests rc=$?"; tail -3 $O/tests_dist.txt
timeout 900 python tools/rank_replay.py --worlds 8 --models sag > $O/replay8.json 2> $O/replay8.err; tail -2 $O/replay8.err
python -c "
import json; d=json.load(open('$O/replay8.json'))
for r in d['replays']: print(r['world'], r['model'], 'wall %.1f chain %.1f' % (r['max_rank_wall_ms'], r['owner_chain_path_ms']), 'unpack', [round(x['unpack_ms'],1) for x in r['ranks']], r['all_owned_columns_bitwise_equal'])"
