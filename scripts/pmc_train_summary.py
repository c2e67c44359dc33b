#!/usr/bin/env python3
"""Summarise scripts/pmc_train.sh: per kernel of the training step, the matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES over
GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), LDS activity and HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE KiB, the gfx950 correction).
usage: pmc_train_summary.py gpurun_out/pmc_train profiles/r02_v5_train_pmc.json"""
import collections, csv, glob, json, sys
src, dst = sys.argv[1], sys.argv[2]
def mean_by_kernel(pattern):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "ww" not in name: continue
            name = name.split("(")[0].replace("void ", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
out = {"source": "scripts/pmc_train.sh: separate rocprofv3 --pmc passes over scripts/bench_train.py (2-conv batch 4096, 3-conv batch 2048), MI355X"}
for arch in ("simple", "full"):
    sq, fe, wr = (mean_by_kernel(f"{src}/{arch}_{x}/**/*counter_collection.csv") for x in ("sq", "fetch", "write"))
    res = {}
    for k in sorted(sq):
        d = {c: sum(v) / len(v) for c, v in sq[k].items()}
        e = {"launches": len(next(iter(sq[k].values())))}
        if d.get("GRBM_GUI_ACTIVE"):
            e["mfma_pipe_busy_frac"] = round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
            e["gpu_cycles"] = d["GRBM_GUI_ACTIVE"] / 8.0
        if d.get("SQ_BUSY_CU_CYCLES"):
            e["lds_active_frac_of_cu_busy"] = round(d.get("SQ_LDS_IDX_ACTIVE", 0.0) / d["SQ_BUSY_CU_CYCLES"], 4)
            e["lds_bank_conflict_frac_of_lds"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, d.get("SQ_LDS_IDX_ACTIVE", 0.0)), 4)
        f = fe.get(k, {}).get("FETCH_SIZE"); w = wr.get(k, {}).get("WRITE_SIZE")
        if f and w: e["hbm_MB_per_launch"] = round((2.0 * sum(f) / len(f) + sum(w) / len(w)) * 1024.0 / 1e6, 2)
        res[k] = e
    out[arch] = res
json.dump(out, open(dst, "w"), indent=1)
for arch in ("simple", "full"):
    print(arch)
    for k, e in out[arch].items():
        if e.get("mfma_pipe_busy_frac", 0) > 0.01 or "grad" in k: print("  %-46s" % k[:46], e)
