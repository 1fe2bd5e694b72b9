#!/bin/bash
# kernel trace of ONE replayed rank (tools/rank_replay.py): what sits on the owners' path per panel.  args: rank_replay arguments
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=gpurun_out/r06p; mkdir -p $R/$O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof -o rp -- python $R/tools/rank_replay.py "$@" > $R/$O/replay.json 2> $R/$O/replay.err )
python tools/rocpd_stats.py $(find $O/prof -name "*.db" | head -1) $O/replay_kernel_stats.txt > /dev/null 2>&1; rm -rf $O/prof
head -24 $O/replay_kernel_stats.txt | cut -c1-150; tail -1 $O/replay.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06p/replay.json") if l.startswith("{")][-1])
for r in d["replays"]:
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, (list, dict))})
    for x in r["ranks"]: print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in x.items() if k.endswith("_ms") or k in ("rank", "owned_panels")})
PY
