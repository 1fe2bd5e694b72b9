#!/bin/bash
set -u
O=gpurun_out/r05i; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for e in "" "PYIPM_NO_CS2=1"; do
for v in "--slices 0" "--slices 1"; do
  env $e timeout 600 python tools/rank_replay.py --nb 256 --worlds 8 --models sag --ranks 1,5 --steps 3 $v > $O/replay_tmp.json 2> $O/replay_tmp.err
  python - "$e $v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r05i/replay_tmp.json"))
for r in d["replays"]:
    for x in r["ranks"]:
        print(sys.argv[1], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in x.items() if k in ("rank", "wall_ms", "factor_ms", "chain_ms", "bcast_ms_incl_link_model", "unpack_ms", "sweeps_ms", "bulk_update_ms", "rows_behind_the_chain_ms", "pack_ms")})
PY
done
done
