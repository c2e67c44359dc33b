# usage (on the GPU box, through gpurun): bash scripts/profile_round.sh <tag>   -> gpurun_out/<tag>/ ; then
#        python scripts/pmc_summary.py gpurun_out/<tag> profiles/rNN_<tag>   (here, after the merge)
set -e
TAG=${1:-v1}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/tests.log 2>&1
timeout -k 10 400 python bench.py > gpurun_out/$TAG/bench.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/stats -o $TAG --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-streaming > $GRAFT_REPO_ROOT/gpurun_out/$TAG/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/stats_timed -o ${TAG}_timed --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --sustained-s 0 --no-cpu-baseline --no-streaming > $GRAFT_REPO_ROOT/gpurun_out/$TAG/stats_timed.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc_fetch -o f --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --sustained-s 0 --no-cpu-baseline --no-streaming > $GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc_write -o w --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --sustained-s 0 --no-cpu-baseline --no-streaming > $GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc_sq -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --sustained-s 0 --no-cpu-baseline --no-streaming > $GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc_sq.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/$TAG/tests.log
tail -1 gpurun_out/$TAG/bench.log | cut -c1-600
