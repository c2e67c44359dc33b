set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
python scripts/dbg_train_cmp.py gpurun_out/r3d/new.npz
WW_LIB_OVERRIDE=$GRAFT_REPO_ROOT/wakeword-jupyterlab_amd/csrc/build/ab/lib_oldcnn.so python scripts/dbg_train_cmp.py gpurun_out/r3d/old.npz
