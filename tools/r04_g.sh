#!/bin/bash
set -u
O=gpurun_out/r04g; mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_merit.py tests/test_gpu_dist.py -q --durations=5 > $O/tests1.txt 2>&1
echo "tests1 rc=$?"; tail -8 $O/tests1.txt
timeout 900 python tools/rank_replay.py > $O/replay_N32768.json 2> $O/replay_N32768.err; echo "replay rc=$?"; tail -8 $O/replay_N32768.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04g/replay_N32768.json"))
print(d["recording"], d["sleep_cycles_per_us"])
for r in d["replays"]:
    print(r["world"], r["model"], "max wall %.1f chain path %.1f implied %.1f bitwise %s" % (r["max_rank_wall_ms"], r["owner_chain_path_ms"], r["implied_step_ms"], r["all_owned_columns_bitwise_equal"]))
    for x in r["ranks"][:2]: print("   ", {k: (round(v,2) if isinstance(v,float) else v) for k,v in x.items()})
PY
for o in "" "--opt dist_head_split=0"; do timeout 600 python bench.py --no-cpu-baseline --config4 off --force-dist --nb 1024 --steps 4 --warmup 2 $o 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('forcedist nb1024 $o', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d.get('dist_phases_per_step',{}).items()})"; done
