#!/bin/bash
# kernel durations of the solve phase with the one-launch backward sweep (rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "--n 3072 --me 768 --mi 1152" ""; do
  rm -rf /tmp/sp; rocprofv3 --kernel-trace --stats -d /tmp/sp -o sp -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline $cfg > /tmp/sp.json 2>/dev/null
  python -c "import json; print(json.load(open('/tmp/sp.json'))['phases_ms_per_step'])"
  python $R/tools/rocpd_stats.py $(find /tmp/sp -name "*.db" | head -1) | grep -E "bwd|fwd|diag_apply|copy_flip|KERNEL"
done
