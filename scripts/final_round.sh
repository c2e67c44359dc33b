# usage (through gpurun): bash scripts/final_round.sh <tag>: profile round + soaks + a 2-rank gloo rehearsal of the bench line
set -e
TAG=${1:-v2}
cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh $TAG
O=gpurun_out/$TAG
PYTHONPATH=. timeout -k 10 120 python scripts/soak.py --what forward --seconds 45 > $O/soak_forward.json 2> $O/soak_forward.err || { tail -5 $O/soak_forward.err; echo SOAK_FWD_FAILED; }
tail -1 $O/soak_forward.json | cut -c1-300
PYTHONPATH=. timeout -k 10 120 python scripts/soak.py --what train --seconds 30 > $O/soak_train.json 2> $O/soak_train.err || { tail -5 $O/soak_train.err; echo SOAK_TRAIN_FAILED; }
tail -1 $O/soak_train.json | cut -c1-300
WW_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-streaming --sustained-s 0.3 > $O/bench_gloo2.log 2> $O/bench_gloo2.err || { tail -8 $O/bench_gloo2.err; echo GLOO2_FAILED; }
grep "^{" $O/bench_gloo2.log | tail -1 | cut -c1-700
