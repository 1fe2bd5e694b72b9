#!/bin/bash
set -u
O=gpurun_out/r05k; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 2400 python tools/rank_replay.py --nvar 65536 --neq 0 --nineq 32768 --host-record --steps 1 --worlds 8 --models sag > $O/replay_N131072_w8.json 2> $O/replay_N131072_w8.err; tail -4 $O/replay_N131072_w8.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05k/replay_N131072_w8.json"))
for r in d["replays"]:
    print("lb %.1f chain %.1f implied %.1f upper %.1f" % (r["max_rank_wall_ms"], r["owner_chain_path_ms"], r["implied_step_ms"], r["implied_step_upper_ms"]), r.get("slice1_chain_path_ms"), r.get("panel_message_path_ms_over_3"))
    for x in r["ranks"][:3]:
        print("   ", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in x.items() if k in ("rank", "wall_ms", "factor_ms", "chain_ms", "bcast_ms_incl_link_model", "unpack_ms", "sweeps_ms", "bulk_update_ms", "rows_behind_the_chain_ms", "pack_ms")})
PY
