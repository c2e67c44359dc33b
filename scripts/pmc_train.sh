# PMC passes over the training step (on the GPU box): matrix-pipe busy fraction and HBM bytes per kernel
#   bash scripts/pmc_train.sh   -> gpurun_out/pmc_train/{simple,full}_{sq,fetch,write}/...; summarise with scripts/pmc_train_summary.py
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_train
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for arch in simple full; do
  b=4096; [ $arch = full ] && b=2048
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY -d $OUT/${arch}_sq -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_train.py $arch $b > $OUT/${arch}_sq.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/${arch}_fetch -o f --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_train.py $arch $b > $OUT/${arch}_fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/${arch}_write -o w --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_train.py $arch $b > $OUT/${arch}_write.log 2>&1
done
echo done
