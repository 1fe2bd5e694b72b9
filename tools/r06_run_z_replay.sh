#!/bin/bash
# the rank replays of the round-6 evidence set (gpurun_out/r06z/replay_*): VERDICT r5 item 2c -- all 8 ranks at N = 131072, a bitwise
# check at nb = 1024 on a size whose recording stays on the device (N = 65536), the implied step at 50 / 75 / 150 GB/s per link
set -u
O=gpurun_out/r06z; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
t0=$(date +%s)
timeout 1500 python tools/rank_replay.py --nb 256 --worlds 2,4,8 --models sag > $O/replay_N32768.json 2> $O/replay_N32768.err; tail -3 $O/replay_N32768.err; echo "t=$(( $(date +%s) - t0 ))"
for L in 50 150; do
  timeout 900 python tools/rank_replay.py --nb 256 --worlds 8 --models sag --link-gbs $L > $O/replay_N32768_link$L.json 2> /dev/null; echo "t=$(( $(date +%s) - t0 ))"
done
timeout 600 python tools/rank_replay.py --nb 256 --worlds 8 --models sag --opt tile_chain=0 > $O/replay_N32768_tile_chain0.json 2> /dev/null; echo "t=$(( $(date +%s) - t0 ))"
timeout 1500 python tools/rank_replay.py --nvar 32768 --neq 8192 --nineq 12288 --nb 1024 --steps 1 --worlds 8 --models sag > $O/replay_N65536_bitwise.json 2> $O/replay_N65536.err; tail -3 $O/replay_N65536.err; echo "t=$(( $(date +%s) - t0 ))"
for L in 75 50 150; do
  timeout 3000 python tools/rank_replay.py --nvar 65536 --neq 0 --nineq 32768 --host-record --steps 1 --worlds 8 --models sag --link-gbs $L > $O/replay_N131072_link$L.json 2> $O/replay_N131072_link$L.err; tail -3 $O/replay_N131072_link$L.err; echo "t=$(( $(date +%s) - t0 ))"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06z/replay_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    for r in d.get("replays", []):
        print(f.split("/")[-1], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, (list, dict))})
PY
