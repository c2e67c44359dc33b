set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v8
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/v8/tests.log 2>&1
timeout -k 10 300 python bench.py > gpurun_out/v8/bench.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/v8/stats -o v8 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-streaming > $GRAFT_REPO_ROOT/gpurun_out/v8/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/v8/pmc_fetch -o f --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --no-cpu-baseline --no-streaming > $GRAFT_REPO_ROOT/gpurun_out/v8/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/v8/pmc_write -o w --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --no-cpu-baseline --no-streaming > $GRAFT_REPO_ROOT/gpurun_out/v8/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY -d $GRAFT_REPO_ROOT/gpurun_out/v8/pmc_sq -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --no-cpu-baseline --no-streaming > $GRAFT_REPO_ROOT/gpurun_out/v8/pmc_sq.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/v8/tests.log
tail -1 gpurun_out/v8/bench.log | cut -c1-600
