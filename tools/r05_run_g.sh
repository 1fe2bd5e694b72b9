#!/bin/bash
set -u
O=gpurun_out/r05g; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
( timeout 900 python -m pytest tests/test_gpu_dist.py -q -x --timeout 420 -k "exchange_forms or selftest or sharing_one_gpu_ragged or drivers_agree or wide_panels" > $O/pytest_dist.log 2>&1; echo "pytest rc=$?" >> $O/pytest_dist.log ); tail -5 $O/pytest_dist.log
for v in "--slices 1 --serialize 0" "--slices 1 --serialize 1" "--slices 0" "--slices 1 --serialize 0"; do
  timeout 600 python tools/rank_replay.py --nb 256 --worlds 8 --models sag --ranks 1,5 --steps 3 $v > $O/replay_tmp.json 2> $O/replay_tmp.err
  python - "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r05g/replay_tmp.json"))
for r in d["replays"]:
    for x in r["ranks"]:
        print(sys.argv[1], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in x.items() if k in ("rank", "wall_ms", "factor_ms", "chain_ms", "bcast_ms_incl_link_model", "unpack_ms", "sweeps_ms", "bulk_update_ms", "rows_behind_the_chain_ms", "pack_ms")})
PY
done
