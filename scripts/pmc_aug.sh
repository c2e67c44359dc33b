# usage (through gpurun): bash scripts/pmc_aug.sh [lib.so]: SQ / LDS / TA counters of the augmentation kernels (counter passes only, no tracing domains)
cd $GRAFT_REPO_ROOT
if [ -n "$1" ]; then export WW_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$1; fi
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_aug
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $O/sq -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_augment.py --check 0 --steps 3 > $O/sq.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM -d $O/sq2 -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_augment.py --check 0 --steps 3 > $O/sq2.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr -d $O/ta -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_augment.py --check 0 --steps 3 > $O/ta.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for d in ("sq", "sq2", "ta"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"gpurun_out/pmc_aug/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("ww::", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k in acc:
        if "resample" in k or "stft" in k or "istft" in k:
            print(d, k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
