#!/bin/bash
# PMC passes over one stage (counters only, no tracing domains).  usage: pmc_passes.sh <stage> <outdir-under-gpurun_out>
set -e
stage=$1; out=$GRAFT_REPO_ROOT/gpurun_out/$2; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line -d $out/p$i -o p$i --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_kernel.py $stage 6 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
done <<'LIST'
SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS
TA_TA_BUSY_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INST_LEVEL_LDS
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_CYCLES
LIST
python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:36]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "ww::" not in k: continue
    print(k)
    for c, vals in sorted(v.items()):
        vals = vals[1:] or vals
        print("   %-40s %16.1f  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
PY
