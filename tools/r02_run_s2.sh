#!/bin/bash
set -u
O=gpurun_out/r02s; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/$O/prof -o fd -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 --force-dist --nb 512 > $R/$O/fd.json 2> $R/$O/fd.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/chain_timeline.py $DB 1500 $O/fd_timeline.txt
rm -rf $O/prof
python - <<'PY'
rows=[l.split() for l in open("gpurun_out/r02s/fd_timeline.txt") if l[0]!='#' and 'start_us' not in l]
# last step: find last k_assemble
idx=max(i for i,r in enumerate(rows) if r[4].startswith('k_assemble'))
rows=rows[idx:]
t0=float(rows[0][0])
fw=[(float(r[0])-t0, float(r[1]), r[4]) for r in rows if r[4] in ('k_fwd_diag','k_fwd_gemv','k_diag_apply')]
ts=[(float(r[0])-t0, float(r[1])) for r in rows if r[4]=='k_tile_step']
print("step kernels", len(rows), "fwd kernels", len(fw), "tile steps", len(ts))
print("last tile_step ends at %.1f us" % (ts[-1][0]+ts[-1][1]))
print("fwd_diag starts (every 8th):", [round(x[0]) for x in fw if x[2]=='k_fwd_diag'][::8])
print("tile_step starts (every 64th):", [round(x[0]) for x in ts][::64])
print("last fwd kernel ends at %.1f" % (fw[-1][0]+fw[-1][1]))
bw=[(float(r[0])-t0) for r in rows if r[4].startswith('k_bwd')]
print("first bwd at", bw[0] if bw else None)
PY
