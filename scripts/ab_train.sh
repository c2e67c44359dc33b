cd $GRAFT_REPO_ROOT
for L in "" wakeword-jupyterlab_amd/csrc/build/ab/lib_nonop.so; do
WW_LIB_OVERRIDE=${L:+$GRAFT_REPO_ROOT/$L} PYTHONPATH=. python - <<PY
import sys, os; sys.path.insert(0,"scripts")
import bench_train
r=bench_train.measure(batch=4096, steps=20, device=0, cpu_sample=2)
print(os.environ.get("WW_LIB_OVERRIDE") or "main", "train simple", round(r["ms_per_step"],4))
PY
done
