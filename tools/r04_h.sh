#!/bin/bash
set -u
O=gpurun_out/r04h; mkdir -p $O
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for r in d["replays"]:
    ch=sum(x["chain_ms"] for x in r["ranks"]); un=sum(x["unpack_ms"] for x in r["ranks"])/len(r["ranks"])
    print(r["world"], r["model"], "max wall %.1f  chain path %.1f  implied %.1f  (sum chain %.1f, mean unpack %.1f, bulk r0 %.1f ms @ %.1f TF/s) bitwise %s" % (
        r["max_rank_wall_ms"], r["owner_chain_path_ms"], r["implied_step_ms"], ch, un, r["ranks"][0]["bulk_update_ms"], r["ranks"][0]["bulk_update_tflops"], r["all_owned_columns_bitwise_equal"]))
PY
}
for v in "persist_rows=1000000" "persist_rows=1000000 --opt reserve_cus=32" "persist_rows=1000000 --opt reserve_cus=64"; do
  tag=$(echo "$v" | tr -c 'a-zA-Z0-9' '_')
  timeout 900 python tools/rank_replay.py --worlds 4,8 --models sag --opt $v > $O/replay_$tag.json 2> $O/replay_$tag.err; echo "== $v rc=$?"; summ $O/replay_$tag.json
done
timeout 1500 python tools/rank_replay.py --nvar 65536 --neq 0 --nineq 32768 --host-record --steps 1 > $O/replay_N131072.json 2> $O/replay_N131072.err; echo "== N131072 rc=$?"; tail -3 $O/replay_N131072.err; summ $O/replay_N131072.json
