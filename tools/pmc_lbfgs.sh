#!/bin/bash
# HBM-side traffic (PMC) of the L-BFGS direction's kernels: the Gram launch (k_update<128,true,8>), the two passes over J
# (k_tall_tn, k_jvec).  Two passes (FETCH_SIZE and WRITE_SIZE do not fit one).  Usage: tools/pmc_lbfgs.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_lbfgs}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
N=262144; ME=1024; MI=3072; M=8; REPS=2
run() { name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "k_update<128, true|k_tall_tn|k_jvec" --output-format csv -d $ROOT/$OUT/$name -o $name -- \
    python $ROOT/tools/bench_lbfgs.py --n $N --me $ME --mi $MI --m $M --reps $REPS --cpu-n 0 > $ROOT/$OUT/$name.log 2>&1
}
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
cd $ROOT
python - "$OUT" $N $ME $MI $M $REPS <<'PY'
import csv, collections, json, sys
src = sys.argv[1]
n, me, mi, m, reps = (int(v) for v in sys.argv[2:7])
p = me + mi
p_pad = (p + 127) // 128 * 128
rr = 2 * m + 1
rows = {nm: list(csv.DictReader(open("%s/%s/%s_counter_collection.csv" % (src, nm, nm)))) for nm in ("tcc1", "tcc2")}
def short(k):
    return k.split("(")[0].replace("pyipm::", "").replace("void ", "")
# the Gram launch is the k_update<128,true,8> dispatch with the largest grid; the factorisation's launches are tiny
out = {}
for key in ("k_tall_tn", "k_jvec", "k_update<128, true, 8>"):
    per = {}
    for nm, cname in (("tcc1", "FETCH_SIZE"), ("tcc2", "WRITE_SIZE")):
        vals = [float(r["Counter_Value"]) for r in rows[nm]
                if (short(r["Kernel_Name"]) == key or (key.startswith("k_update") and short(r["Kernel_Name"]).startswith(key)))
                and r["Counter_Name"] == cname]
        if key.startswith("k_update"):
            top = max(vals) if vals else 0.0
            vals = [v for v in vals if v > 0.5 * top]          # the Gram launches only
        per[cname] = (sum(vals) / len(vals) if vals else 0.0, len(vals))
    fetch = 2.0 * 1024.0 * per["FETCH_SIZE"][0]                # KB; doubled on gfx950 (MI355X_MICROARCH.md)
    write = 1024.0 * per["WRITE_SIZE"][0]
    out[key] = {"launches_counted": per["FETCH_SIZE"][1], "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write}
J = 8.0 * n * p_pad
out["k_tall_tn"]["algorithmic_bytes_per_launch"] = J + 8.0 * n * rr
out["k_jvec"]["algorithmic_bytes_per_launch"] = J + 8.0 * n
out["k_update<128, true, 8>"]["algorithmic_bytes_per_launch"] = J            # one read of the operand; C stays in registers
nt = p_pad // 128
out["k_update<128, true, 8>"]["operand_bytes_without_cache_reuse"] = (nt * (nt + 1) // 2) * 2 * 128 * 8.0 * n   # every tile streams its two 128-row panels
for k, v in out.items():
    v["traffic_over_algorithmic"] = (v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"]) / v["algorithmic_bytes_per_launch"]
json.dump({"workload": "tools/bench_lbfgs.py --n %d --me %d --mi %d --m %d" % (n, me, mi, m),
           "units": "FETCH_SIZE x2 (gfx950), KB -> bytes; averages per launch", "kernels": out}, open("%s/pmc_lbfgs.json" % src, "w"), indent=1)
for k, v in out.items():
    print(k, v)
PY
