#!/bin/bash
# round 6: receivers rebuild L straight from the panel message (no copy of W first) -- the distributed tests, then the replays
set -u
O=gpurun_out/r06x; mkdir -p $O
( timeout 2400 python -m pytest tests/test_gpu_dist.py -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
timeout 3000 python tools/rank_replay.py --nvar 65536 --neq 0 --nineq 32768 --host-record --steps 1 --worlds 8 --models sag --serialize 1 > $O/replay_N131072_one_comm_fused_unpack.json 2> $O/replay.err; tail -1 $O/replay.err
timeout 1500 python tools/rank_replay.py --nvar 32768 --neq 8192 --nineq 12288 --nb 1024 --steps 1 --worlds 8 --models sag --serialize 1 > $O/replay_N65536_one_comm_fused_unpack.json 2> $O/replay.err; tail -1 $O/replay.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06x/replay_*fused_unpack.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    for r in d["replays"]:
        print(f.split('/')[-1], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, (list, dict))})
        print('   unpack_ms per rank', [round(x["unpack_ms"], 1) for x in r["ranks"]], 'wall', [round(x["wall_ms"], 1) for x in r["ranks"]])
PY
