#!/bin/bash
set -u
O=gpurun_out/r05j; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
( timeout 1200 python -m pytest tests/test_gpu_dist.py -q -x --timeout 420 -k "exchange_forms or selftest or sharing_one_gpu or drivers_agree or wide_panels" > $O/pytest_dist.log 2>&1; echo "pytest rc=$?" >> $O/pytest_dist.log ); tail -5 $O/pytest_dist.log
for v in "--slices 1" "--slices 0" "--slices 1 --serialize 1"; do
  timeout 900 python tools/rank_replay.py --nb 256 --worlds 8 --models sag --steps 3 $v > $O/replay_tmp.json 2> $O/replay_tmp.err
  cp $O/replay_tmp.json "$O/replay_$(echo $v | tr -d ' -').json"
  python - "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r05j/replay_tmp.json"))
for r in d["replays"]:
    print(sys.argv[1], "lb %.1f chain %.1f implied %.1f upper %.1f" % (r["max_rank_wall_ms"], r["owner_chain_path_ms"], r["implied_step_ms"], r["implied_step_upper_ms"]), r.get("slice1_chain_path_ms"), r.get("panel_message_path_ms_over_3"), r["all_owned_columns_bitwise_equal"])
    for x in r["ranks"][:3]:
        print("   ", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in x.items() if k in ("rank", "wall_ms", "factor_ms", "chain_ms", "bcast_ms_incl_link_model", "unpack_ms", "sweeps_ms", "bulk_update_ms", "rows_behind_the_chain_ms", "pack_ms")})
PY
done
