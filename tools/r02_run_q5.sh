#!/bin/bash
set -u
O=gpurun_out/r02q; mkdir -p $O
B="timeout 600 python bench.py --no-cpu-baseline"
for v in "a:" "b:--opt head_serial=1" "c:--opt head_on_side=0" "d:--opt group_chain=0 --opt head_on_side=0"; do
  k=${v%%:*}; o=${v#*:}
  $B --steps 5 --warmup 2 $o > $O/m_$k.json 2> $O/m_$k.err
  $B --nvar 2048 --neq 0 --nineq 2048 --steps 20 --warmup 3 $o > $O/c2_$k.json 2> $O/c2_$k.err
  $B --nvar 16384 --neq 8192 --nineq 8192 --steps 3 --warmup 1 $o > $O/c3_$k.json 2> $O/c3_$k.err
done
python - <<'PY'
import json
for c in ("m","c2","c3"):
  for k in "abcd":
    f="%s_%s"%(c,k)
    try:
        d=json.load(open("gpurun_out/r02q/%s.json"%f)); print(f, "%.3f ms"%d["ms_per_step"], {a:round(b,3) for a,b in d["phases_ms_per_step"].items()}, "upd %.1f TF"%d["roofline"]["achieved"], d["backward_error"])
    except Exception as e:
        print(f,"FAILED",e, open("gpurun_out/r02q/%s.err"%f).read()[-800:])
PY
