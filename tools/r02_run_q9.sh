#!/bin/bash
set -u
O=gpurun_out/r02q; mkdir -p $O
run() { # name cfg opts env
  env $4 timeout 600 python bench.py --no-cpu-baseline $2 $3 > $O/t_$1.json 2> $O/t_$1.err
  python - $1 <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/r02q/t_%s.json" % sys.argv[1])); print(sys.argv[1], "%.3f ms" % d["ms_per_step"], {k: round(v, 3) for k, v in d["phases_ms_per_step"].items()}, d["backward_error"])
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/r02q/t_%s.err" % sys.argv[1]).read()[-400:])
PY
}
C2="--nvar 2048 --neq 0 --nineq 2048 --steps 20 --warmup 3"
M="--steps 5 --warmup 2"
run c2_base "$C2" "--opt tail_group=4" "X=1"
run c2_g8 "$C2" "--opt tail_group=8" "PYIPM_NEWTON_GROUP=8"
run c2_noearly "$C2" "--opt tail_group=4 --opt early_head=0" "X=1"
run c2_g6 "$C2" "--opt tail_group=6" "PYIPM_NEWTON_GROUP=6"
run m_base "$M" "--opt tail_group=4" "X=1"
run m_g8 "$M" "--opt tail_group=8" "PYIPM_NEWTON_GROUP=8"
run m_g8t4 "$M" "--opt tail_group=4" "PYIPM_NEWTON_GROUP=8"
run m_noearly "$M" "--opt tail_group=4 --opt early_head=0" "X=1"
run m_base2 "$M" "--opt tail_group=4" "X=1"
