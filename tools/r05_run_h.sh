#!/bin/bash
set -u
O=gpurun_out/r05h; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
B="timeout 600 python bench.py --no-cpu-baseline --config4 off --configs off --steps 8 --warmup 2 --no-clock"
for rep in 1 2; do
  for q in 4 8 16; do
    GPU_MAX_HW_QUEUES=$q $B > $O/bench_q${q}_$rep.json 2> /dev/null
    GPU_MAX_HW_QUEUES=$q $B --nvar 2048 --neq 0 --nineq 2048 --steps 30 --warmup 5 > $O/cfg2_q${q}_$rep.json 2> /dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05h/*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], "%.3f ms" % d["ms_per_step"], "panel %.2f" % d["phases_ms_per_step"]["panel(tile+scale+in-panel)"], "trailing %.2f" % d["phases_ms_per_step"]["trailing"])
    except Exception as e:
        print(f, "FAILED", e)
PY
for q in 4 8; do
for v in "--slices 0" "--slices 1"; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python tools/rank_replay.py --nb 256 --worlds 8 --models sag --ranks 1,5 --steps 3 $v > $O/replay_tmp.json 2> $O/replay_tmp.err
  python - "q=$q $v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r05h/replay_tmp.json"))
for r in d["replays"]:
    for x in r["ranks"]:
        print(sys.argv[1], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in x.items() if k in ("rank", "wall_ms", "factor_ms", "chain_ms", "bcast_ms_incl_link_model", "unpack_ms", "sweeps_ms", "bulk_update_ms", "rows_behind_the_chain_ms", "pack_ms")})
PY
done
done
