#!/bin/bash
# round 6: the second slice's rows inside the chain's launch (dist_slices = 2), wide panels: N = 131072 on the wire form of one communicator
set -u
O=gpurun_out/r06x; mkdir -p $O
for sl in 1 2; do
  timeout 3000 python tools/rank_replay.py --nvar 65536 --neq 0 --nineq 32768 --host-record --steps 1 --worlds 8 --models sag --serialize 1 --slices $sl > $O/replay_N131072_one_comm_slices$sl.json 2> $O/replay.err; tail -1 $O/replay.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06x/replay_N131072_one_comm_slices*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    for r in d["replays"]:
        print(f.split('/')[-1], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, (list, dict))})
        print('   chain_ms per rank', [round(x["chain_ms"], 2) for x in r["ranks"]], 'wall', [round(x["wall_ms"], 1) for x in r["ranks"]])
PY
