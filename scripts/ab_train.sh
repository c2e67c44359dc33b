# A/B of the training step (2-conv and 3-conv) across builds of the library, one subprocess per build, two interleaved rounds.
# usage (through gpurun): bash scripts/ab_train.sh [lib_a.so lib_b.so ...]   ("" = the shipped library)
cd $GRAFT_REPO_ROOT
LIBS=("" "$@")
for rnd in 1 2; do
for L in "${LIBS[@]}"; do
WW_LIB_OVERRIDE=${L:+$GRAFT_REPO_ROOT/$L} PYTHONPATH=. python - <<PY
import sys, os; sys.path.insert(0,"scripts")
import bench_train
r=bench_train.measure(batch=4096, steps=20, device=0, cpu_sample=2)
f=bench_train.measure(batch=2048, steps=10, device=0, arch="full", cpu_sample=2)
print(os.path.basename(os.environ.get("WW_LIB_OVERRIDE") or "main"), "train simple", round(r["ms_per_step"],4), "ms; full (2048)", round(f["ms_per_step"],4), "ms", flush=True)
PY
done
done
