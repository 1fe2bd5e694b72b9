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
for s in "8k:--nvar 4096 --neq 0 --nineq 2048" "12k:--nvar 6144 --neq 0 --nineq 3072" "16k:--nvar 8192 --neq 0 --nineq 4096" "3k:--nvar 1024 --neq 0 --nineq 1024"; do
  k=${s%%:*}; c=${s#*:}
  run ${k}_g4 "$c --steps 10 --warmup 3" "--opt tail_group=4" "X=1"
  run ${k}_g8 "$c --steps 10 --warmup 3" "--opt tail_group=8" "PYIPM_NEWTON_GROUP=8"
  run ${k}_g8t4 "$c --steps 10 --warmup 3" "--opt tail_group=4" "PYIPM_NEWTON_GROUP=8"
done
