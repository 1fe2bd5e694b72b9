#!/bin/bash
set -u
O=gpurun_out/r05l; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { name=$1; shift
  timeout 2400 python tools/rank_replay.py --nvar 65536 --neq 0 --nineq 32768 --host-record --steps 1 --worlds 8 --models sag --ranks 0,1,2 "$@" > $O/replay_$name.json 2> $O/replay_$name.err; tail -2 $O/replay_$name.err
  python - $name <<'PY'
import json, sys
d = json.load(open("gpurun_out/r05l/replay_%s.json" % sys.argv[1]))
for r in d["replays"]:
    print(sys.argv[1], "lb %.1f chain(3 ranks) %.1f" % (r["max_rank_wall_ms"], r["owner_chain_path_ms"]), r.get("slice1_chain_path_ms"), r.get("panel_message_path_ms_over_3"))
    for x in r["ranks"][:3]:
        print("   ", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in x.items() if k in ("rank", "wall_ms", "factor_ms", "chain_ms", "bcast_ms_incl_link_model", "unpack_ms", "sweeps_ms", "bulk_update_ms", "rows_behind_the_chain_ms", "pack_ms")})
PY
}
run persist --opt persist_rows=1048576
run nb512 --nb 512
run nb512_persist --nb 512 --opt persist_rows=1048576
