#!/bin/bash
# SQ / LDS counter passes over the conv kernel alone (counters only, no tracing domains).  usage: pmc_k2.sh <outdir-under-gpurun_out> [env...]
#   e.g.  bash scripts/pmc_k2.sh k2_w16        bash scripts/pmc_k2.sh k2_x32 WW_K2_FORM=x32
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $out
for kv in "$@"; do export "$kv"; done
cd /tmp; export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line -d $out/p$i -o p$i --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_kernel.py cnn 6 > $out/p$i.log 2>&1 || { echo "pass $i failed: $line"; tail -3 $out/p$i.log; }
done <<'LIST'
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES
LIST
python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1] + "/summary.txt", "w") as o:
    for k, v in agg.items():
        if "ww::cnn" not in k: continue
        print(k, file=o)
        for c, vals in sorted(v.items()):
            vals = vals[1:] or vals
            print("   %-40s %16.1f  (n=%d)" % (c, sum(vals) / len(vals), len(vals)), file=o)
print(open(sys.argv[1] + "/summary.txt").read())
PY
