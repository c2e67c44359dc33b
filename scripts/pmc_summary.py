#!/usr/bin/env python3
"""Summarise the PMC passes of scripts/profile_round.sh into profiles/<tag>_pmc_traffic.json (read by bench.py's
roofline.traffic).  usage: pmc_summary.py <gpurun_out/vN dir> <profiles/rNN_vN prefix>

HBM bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE, both reported in KiB (the gfx950 correction of MI355X_MICROARCH.md,
HBM / rocprofv3 section); passes are separate (--pmc only, no tracing domains)."""
import collections
import csv
import glob
import json
import shutil
import sys

src, prefix = sys.argv[1], sys.argv[2]


def mean_by_kernel(pattern):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "ww::" not in name:
                continue
            name = name[name.index("ww::"):].split("(")[0]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


fetch = mean_by_kernel(src + "/pmc_fetch/**/*counter_collection.csv")
write = mean_by_kernel(src + "/pmc_write/**/*counter_collection.csv")
sq = mean_by_kernel(src + "/pmc_sq/**/*counter_collection.csv")
out = {"source": "scripts/profile_round.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ set as separate passes over "
                 "`python bench.py --steps 5 --no-cpu-baseline --no-streaming` (batch 4096, arch simple, default conv math), MI355X",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction)",
       "kernels": {}, "sq_counters_mean": {}}
for k in sorted(fetch):
    f = fetch[k]["FETCH_SIZE"][1:] or fetch[k]["FETCH_SIZE"]          # drop the first (cold) launch
    w = write.get(k, {}).get("WRITE_SIZE", [0.0])
    w = w[1:] or w
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    out["kernels"][k] = {"FETCH_SIZE_KiB_mean": fm, "WRITE_SIZE_KiB_mean": wm, "launches": len(f),
                         "hbm_bytes_per_launch_corrected": (2.0 * fm + wm) * 1024.0}
for k in sorted(sq):
    out["sq_counters_mean"][k] = {c: sum(v[1:] or v) / len(v[1:] or v) for c, v in sorted(sq[k].items())}
    d = out["sq_counters_mean"][k]
    if d.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        # GRBM_GUI_ACTIVE sums the 8 XCDs; MFMA busy sums 256 CUs x 4 SIMDs
        d["mfma_pipe_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
out["algorithmic_bytes_per_launch"] = {"ww::logmel_kernel<false>": 4096 * 74240, "ww::cnn2w_kernel<true> / cnn2h16_kernel<true>": 4096 * (80 * 32 * 4 + 64 * 4) + 1047040}
json.dump(out, open(prefix + "_pmc_traffic.json", "w"), indent=1)
for name, dst in (("stats/*kernel_stats.csv", "_bench_kernel_stats.csv"), ("pmc_fetch/*counter_collection.csv", "_pmc_fetch_counter_collection.csv"),
                  ("pmc_write/*counter_collection.csv", "_pmc_write_counter_collection.csv"), ("pmc_sq/*counter_collection.csv", "_pmc_sq_counter_collection.csv")):
    for f in glob.glob(src + "/" + name):
        shutil.copy(f, prefix + dst)
line = [x for x in open(src + "/bench.log") if x.startswith("{")][-1]
open(prefix + "_bench_line.json", "w").write(line)
print(json.dumps({k: v["hbm_bytes_per_launch_corrected"] for k, v in out["kernels"].items()}))
print({k: v.get("mfma_pipe_busy_frac") for k, v in out["sq_counters_mean"].items()})
