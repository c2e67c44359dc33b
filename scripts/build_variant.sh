#!/bin/bash
# Build an ablation / experiment twin of the library: scripts/build_variant.sh <name> [extra hipcc flags...]
#   -> wakeword-jupyterlab_amd/csrc/build/ab/lib_<name>.so (travels to the GPU box with gpurun; git-ignored)
# Use with scripts/ab_kernels.py (interleaved A/B in one process) or WW_LIB_OVERRIDE.
set -e
name=$1; shift
cd "$(dirname "$0")/../wakeword-jupyterlab_amd/csrc"
root=$(cd ../.. && pwd)
obj=build/var_$name
mkdir -p $obj build/ab
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$root/include -I. -Wall -Wno-unused-function -fvisibility=hidden -DWW_BUILD -fno-slp-vectorize"
pids=()
for src in ww_tables.cpp ww_logmel.hip ww_cnn.hip ww_head.hip ww_decode.hip ww_files.cpp ww_augment.hip ww_train.hip ww_train_h.hip ww_api.hip; do
  per_file=""                      # the Makefile's per-file scheduling strategies (a variant must differ from the shipped library only in what it is asked to)
  [ $src = ww_cnn.hip ] && per_file="-mllvm -amdgpu-sched-strategy=max-ilp"
  [ $src = ww_logmel.hip ] && per_file="-mllvm -amdgpu-sched-strategy=iterative-maxocc"
  /opt/rocm/bin/hipcc $flags $per_file "$@" -x hip -c $src -o $obj/${src%.*}.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/ab/lib_$name.so $obj/*.o
ls -la build/ab/lib_$name.so
