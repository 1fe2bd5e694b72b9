#!/bin/bash
set -u
O=gpurun_out/r02q; mkdir -p $O
B="timeout 600 python bench.py --no-cpu-baseline"
$B --nvar 2048 --neq 0 --nineq 2048 --steps 20 --warmup 3 > $O/c2_new.json 2> $O/c2_new.err
$B --nvar 2048 --neq 0 --nineq 2048 --steps 20 --warmup 3 --opt group_chain=0 > $O/c2_old.json 2> $O/c2_old.err
$B --steps 5 --warmup 2 > $O/m_new.json 2> $O/m_new.err
$B --steps 5 --warmup 2 --opt group_chain=0 > $O/m_old.json 2> $O/m_old.err
python - <<'PY'
import json
for f in ("c2_new","c2_old","m_new","m_old"):
    try:
        d=json.load(open("gpurun_out/r02q/%s.json"%f)); print(f, "%.3f ms"%d["ms_per_step"], d["phases_ms_per_step"], "upd %.1f TF"%d["roofline"]["achieved"], d["inertia"]["n_neg"], d["backward_error"])
    except Exception as e:
        print(f,"FAILED",e, open("gpurun_out/r02q/%s.err"%f).read()[-800:])
PY
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/$O/prof -o c2 -- python $R/bench.py --no-cpu-baseline --nvar 2048 --neq 0 --nineq 2048 --steps 3 --warmup 2 > $R/$O/c2.json 2> $R/$O/c2.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/c2_stats.txt | head -12
python tools/chain_timeline.py $DB 500 $O/c2_timeline.txt
rm -rf $O/prof
