# per-kernel times of the training step for ablation twins of the library (on the GPU box):
#   bash scripts/prof_train_variants.sh <arch> <variant> [<variant> ...]     ("main" = the shipped library) -> gpurun_out/trainv/<variant>.txt
arch=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/trainv
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = main ]; then unset WW_LIB_OVERRIDE; else export WW_LIB_OVERRIDE=$GRAFT_REPO_ROOT/wakeword-jupyterlab_amd/csrc/build/ab/lib_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/pv_$v -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_train.py $arch > /tmp/pv_$v.log 2>&1 || exit 1
  python3 - "$v" <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/trainv/$v.txt
import csv, sys, glob
v = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(f"/tmp/pv_{v}/**/t_kernel_stats.csv", recursive=True)[0])))
for r in rows[:8]:
    print("%-60s avg %9.1f us x %s" % (r["Name"][:60], float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
  echo "== $v"; head -6 $GRAFT_REPO_ROOT/gpurun_out/trainv/$v.txt
done
