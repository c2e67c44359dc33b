# usage (through gpurun): bash scripts/ab_aug.sh lib1.so lib2.so ...: per-kernel times of the augmentation for several builds of the library
cd $GRAFT_REPO_ROOT
for L in "$@"; do
  export WW_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$L
  (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ab_aug/$(basename $L .so) -o aug --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_augment.py --check 0 > /dev/null 2>&1)
  python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/ab_aug/$(basename $L .so)/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        nm = r["Name"].split("(")[0].replace("ww::", "")
        if nm.split("<")[0].endswith(("resample_kernel", "stft_kernel", "pv_kernel", "istft_kernel", "stft_pv_kernel", "noise_kernel", "roll_kernel")):
            print("$(basename $L)", nm, r["Calls"], round(float(r["AverageNs"])/1e6,4), "ms")
PY
done
