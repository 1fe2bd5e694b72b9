#!/bin/bash
# round 6: the second slice's rows inside the chain's launch (dist_slices = 2) -- the distributed tests, then the replay on the wire form of one communicator
set -u
O=gpurun_out/r06x; mkdir -p $O
( timeout 2400 python -m pytest tests/test_gpu_dist.py tests/test_gpu_tile_blocked.py tests/test_gpu_symmetric.py -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
for sl in 1 2; do
  timeout 900 python tools/rank_replay.py --nb 256 --worlds 2,4,8 --models sag --slices $sl --serialize 1 > $O/replay_N32768_one_comm_slices$sl.json 2> $O/replay.err; tail -1 $O/replay.err
done
timeout 1500 python tools/rank_replay.py --nvar 32768 --neq 8192 --nineq 12288 --nb 256 --steps 1 --worlds 8 --models sag --serialize 1 > $O/replay_N65536_nb256_one_comm.json 2> $O/replay.err; tail -1 $O/replay.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06x/replay_*one_comm*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    for r in d["replays"]:
        print(f.split('/')[-1], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, (list, dict))})
PY
