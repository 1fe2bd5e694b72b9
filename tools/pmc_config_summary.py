#!/usr/bin/env python
"""Per-kernel summary of three rocprofv3 counter passes of one bench.py configuration (tools/r03_run_z.sh: tcc1 = FETCH_SIZE,
tcc2 = WRITE_SIZE, sq1 = MFMA busy): HBM-side bytes (FETCH_SIZE doubled per MI355X_MICROARCH.md, KB -> bytes), MFMA pipe
busy fraction and executed fp64 MFMA flops, per kernel.  usage: pmc_config_summary.py <dir> <out.json>"""
import collections, csv, json, os, sys
src, dst = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(collections.Counter)
calls = collections.Counter()
for name in ("tcc1", "tcc2", "sq1"):
    f = "%s/%s/%s_counter_collection.csv" % (src, name, name)
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("pyipm::", "").replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if name == "tcc1" and r["Counter_Name"] == "FETCH_SIZE":
            calls[k] += 1
out = {}
for k, c in tot.items():
    fetch, write = 2.0 * 1024.0 * c["FETCH_SIZE"], 1024.0 * c["WRITE_SIZE"]
    # as tools/pmc_summary.py: SQ_VALU_MFMA_BUSY_CYCLES over GRBM_GUI_ACTIVE x 128 (kernel cycles from the FETCH pass of the same run)
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128.0) if c["GRBM_GUI_ACTIVE"] else None
    out[k] = {"launches": calls[k], "hbm_read_bytes": fetch, "hbm_write_bytes": write, "GRBM_GUI_ACTIVE": c["GRBM_GUI_ACTIVE"],
              "mfma_busy_frac_of_kernel_cycles": busy, "f64_mfma_flops": 512.0 * c["SQ_INSTS_VALU_MFMA_MOPS_F64"]}
top = sorted(out.items(), key=lambda kv: -(kv[1]["hbm_read_bytes"] + kv[1]["hbm_write_bytes"]))
json.dump({"units": "bytes over all launches of the run (warm-up included); FETCH_SIZE x2 (gfx950), counters in KB", "kernels": dict(top)},
          open(dst, "w"), indent=1)
for k, v in top[:8]:
    print("%-40s launches %5d read %.3e B write %.3e B mfma_busy %s flops %.3e" % (k[:40], v["launches"], v["hbm_read_bytes"], v["hbm_write_bytes"],
                                                                                 "%.2f" % v["mfma_busy_frac_of_kernel_cycles"] if v["mfma_busy_frac_of_kernel_cycles"] is not None else "-", v["f64_mfma_flops"]))
