# usage (on the GPU box, through gpurun): bash scripts/profile_full.sh <tag>  -> gpurun_out/<tag>_full/: the 3-conv WakewordModel's bench line,
# kernel stats and HBM-traffic PMC passes (separate --pmc passes, no tracing domains); summarise with
#   python scripts/pmc_summary.py gpurun_out/<tag>_full profiles/rNN_<tag>_full
set -e
TAG=${1:-v1}_full
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
A="--arch full --no-streaming"
timeout -k 10 400 python bench.py $A > $O/bench.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o $TAG --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $A --no-cpu-baseline --sustained-s 0.5 > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $A --steps 5 --sustained-s 0 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $A --steps 5 --sustained-s 0 --no-cpu-baseline > $O/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY -d $O/pmc_sq -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $A --steps 5 --sustained-s 0 --no-cpu-baseline > $O/pmc_sq.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $O/bench.log | cut -c1-300
