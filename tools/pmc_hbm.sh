#!/bin/bash
# HBM-side traffic (PMC) of the HBM-bound kernels K1 (k_assemble) and K5 (k_fwd_gemv, k_bwd_sweep; k_bwd_dot / k_fwd_sweep where they run): two passes
# (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2).  Usage: tools/pmc_hbm.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_hbm}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --config4 off --configs off --no-clock"
run() { name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "k_assemble|k_fwd_gemv|k_bwd_dot|k_bwd_sweep|k_fwd_sweep|k_symv_tiles" --output-format csv -d $ROOT/$OUT/$name -o $name -- python $ROOT/bench.py $ARGS > $ROOT/$OUT/$name.log 2>&1
}
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
cd $ROOT
python - "$OUT" <<'PY'
import csv, collections, json, sys
src = sys.argv[1]
tot = collections.defaultdict(lambda: collections.Counter())
calls = collections.Counter()
for name in ("tcc1", "tcc2"):
    for r in csv.DictReader(open("%s/%s/%s_counter_collection.csv" % (src, name, name))):
        k = r["Kernel_Name"].split("(")[0].replace("pyipm::", "").replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if name == "tcc1":
            calls[k] += 1
N, n, me, mi, steps = 32768, 16384, 4096, 6144, 2
alg = {"k_assemble": 4.0 * N * N + 8.0 * (n * n / 2.0 + n * me + n * mi),          # per step
       "k_fwd_gemv": 4.0 * N * N, "k_bwd_dot": 4.0 * N * N, "k_bwd_sweep": 4.0 * N * N, "k_fwd_sweep": 4.0 * N * N}
out = {}
for k in tot:
    fetch = 2.0 * 1024.0 * tot[k]["FETCH_SIZE"]          # KB, doubled on gfx950 (MI355X_MICROARCH.md)
    write = 1024.0 * tot[k]["WRITE_SIZE"]
    out[k] = {"launches": calls[k], "hbm_read_bytes_per_step": fetch / steps, "hbm_write_bytes_per_step": write / steps,
              "algorithmic_bytes_per_step": alg.get(k), "traffic_over_algorithmic": (fetch + write) / steps / alg[k] if k in alg else None}
json.dump({"workload": "bench.py --steps 1 --warmup 1 (2 steps), N=32768", "units": "FETCH_SIZE x2 (gfx950), KB -> bytes", "kernels": out},
          open("%s/pmc_hbm.json" % src, "w"), indent=1)
for k, v in out.items():
    print(k, v)
PY
