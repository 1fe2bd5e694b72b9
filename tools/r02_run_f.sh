#!/bin/bash
set -u
O=gpurun_out/r02f; mkdir -p $O
python tools/bench_tile.py > $O/bench_tile.txt 2>&1; python tools/tile_clock.py >> $O/bench_tile.txt 2>&1; cat $O/bench_tile.txt
python - <<'PY' 2>&1 | tee gpurun_out/r02f/hbm_roofline.txt
import torch, time
dev = torch.device("cuda", 0)
n = 1 << 30   # 8 GiB of doubles
a = torch.empty(n, dtype=torch.float64, device=dev); b = torch.empty(n, dtype=torch.float64, device=dev)
def t(f, reps=5):
    f(); torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
ms = t(lambda: a.fill_(1.0)); print("fill  8 GiB: %.3f ms = %.2f TB/s (write only)" % (ms, n * 8 / ms / 1e9))
ms = t(lambda: b.copy_(a)); print("copy  8 GiB: %.3f ms = %.2f TB/s (read + write bytes)" % (ms, 2 * n * 8 / ms / 1e9))
ms = t(lambda: a.sum()); print("sum   8 GiB: %.3f ms = %.2f TB/s (read only)" % (ms, n * 8 / ms / 1e9))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find $O/prof -name "*.db" | head -1) $O/kernel_stats.txt 2>/dev/null | head -32
