#!/bin/bash
set -u
O=gpurun_out/r02s; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_symmetric.py tests/test_gpu_parity.py -m gpu -q > $O/pytest_a.log 2>&1; tail -5 $O/pytest_a.log
B="timeout 600 python bench.py --no-cpu-baseline --steps 4 --warmup 1"
for v in "single:" "fd512:--force-dist --nb 512" "fd512old:--force-dist --nb 512 --opt tile_step=0" "fd256:--force-dist --nb 256"; do
  k=${v%%:*}; o=${v#*:}
  $B $o > $O/$k.json 2> $O/$k.err
done
python - <<'PY'
import json
for k in ("single","fd512","fd512old","fd256"):
    try:
        d=json.load(open("gpurun_out/r02s/%s.json"%k)); print(k, "%.2f ms"%d["ms_per_step"], {a:round(b,2) for a,b in d["phases_ms_per_step"].items()}, d.get("dist_phases_per_step"), d["backward_error"])
    except Exception as e:
        print(k,"FAILED",e, open("gpurun_out/r02s/%s.err"%k).read()[-600:])
PY
