#!/bin/bash
# the rank replays of the round-5 evidence set (gpurun_out/r05z/replay_*), apart from tools/r05_run_z.sh: they take the longest
set -u
O=gpurun_out/r05z; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
t0=$(date +%s)
timeout 1500 python tools/rank_replay.py --nb 256 --worlds 2,4,8 --models sag > $O/replay_N32768.json 2> $O/replay_N32768.err; tail -4 $O/replay_N32768.err; echo "t=$(( $(date +%s) - t0 ))"
timeout 600 python tools/rank_replay.py --nb 256 --worlds 8 --models sag --slices 0 > $O/replay_N32768_one_message.json 2> /dev/null; echo "t=$(( $(date +%s) - t0 ))"
timeout 600 python tools/rank_replay.py --nb 256 --worlds 8 --models sag --opt tile_waves=4 > $O/replay_N32768_tile_waves4.json 2> /dev/null; echo "t=$(( $(date +%s) - t0 ))"
timeout 3000 python tools/rank_replay.py --nvar 65536 --neq 0 --nineq 32768 --host-record --steps 1 --worlds 8 --models sag --ranks 0,1,2 > $O/replay_N131072.json 2> $O/replay_N131072.err; tail -4 $O/replay_N131072.err; echo "t=$(( $(date +%s) - t0 ))"
python - <<'PY'
import json
for f in ("replay_N32768", "replay_N32768_one_message", "replay_N32768_tile_waves4", "replay_N131072"):
    try:
        d = json.loads([l for l in open("gpurun_out/r05z/%s.json" % f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    for r in d.get("replays", []):
        print(f, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, (list, dict))})
PY
