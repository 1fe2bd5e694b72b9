#!/bin/bash
set -u
O=gpurun_out/r05n; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYIPM_EXPERT=1
timeout 2400 python -m pytest tests/test_gpu_dist.py -x -q > $O/pytest_dist.log 2>&1; tail -3 $O/pytest_dist.log
for v in 0 1; do
timeout 900 python tools/rank_replay.py --nb 256 --worlds 8 --models sag --opt dist_unpack_side=$v > $O/replay_u$v.json 2> $O/replay_u$v.err; tail -1 $O/replay_u$v.err
done
for v in 0 1; do
timeout 3000 python tools/rank_replay.py --nvar 65536 --neq 0 --nineq 32768 --host-record --steps 1 --worlds 8 --models sag --ranks 0,1,2 --opt dist_unpack_side=$v > $O/replay_N131072_u$v.json 2> $O/replay_N131072_u$v.err; tail -1 $O/replay_N131072_u$v.err
done
python - <<'PY'
import json
for f in ("replay_u0", "replay_u1", "replay_N131072_u0", "replay_N131072_u1"):
    d = json.loads([l for l in open("gpurun_out/r05n/%s.json" % f) if l.startswith("{")][-1])
    for r in d["replays"]:
        print(f, {k: (round(x, 2) if isinstance(x, float) else x) for k, x in r.items() if not isinstance(x, (list, dict))})
        for x in r["ranks"][:2]:
            print("   ", {k: (round(y, 2) if isinstance(y, float) else y) for k, y in x.items() if k in ("rank", "wall_ms", "factor_ms", "unpack_ms", "bulk_update_ms", "chain_ms", "owned_columns_bitwise_equal_to_recorded_factor")})
PY
