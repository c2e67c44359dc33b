set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
echo "== training test, current library"
timeout -k 10 300 python -m pytest tests/test_gpu_train.py -m gpu -q -k "split_precision_backward" > $O/t_cur.log 2>&1 || true
tail -3 $O/t_cur.log | cut -c1-300
grep -E "AssertionError: conv" $O/t_cur.log | head -3 || true
echo "== training test, round-2 conv kernels"
WW_LIB_OVERRIDE=$GRAFT_REPO_ROOT/wakeword-jupyterlab_amd/csrc/build/ab/lib_oldcnn.so timeout -k 10 300 python -m pytest tests/test_gpu_train.py -m gpu -q -k "split_precision_backward" > $O/t_old.log 2>&1 || true
tail -3 $O/t_old.log | cut -c1-300
grep -E "AssertionError: conv" $O/t_old.log | head -3 || true
echo "== full suite"
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=15 > $O/tests.log 2>&1 || { grep -E "^(FAILED|ERROR)|AssertionError" $O/tests.log | head -40; echo TESTS_FAILED; }
tail -3 $O/tests.log
