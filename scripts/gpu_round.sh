# usage (through gpurun): bash scripts/gpu_round.sh <tag>  -> gpurun_out/<tag>/{tests.log,bench.log}: the GPU test suite and the default bench line
set -e
TAG=${1:-t}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=15 > $O/tests.log 2>&1 || { grep -E "^(FAILED|ERROR)|AssertionError" $O/tests.log | head -40; echo TESTS_FAILED; }
tail -2 $O/tests.log
timeout -k 10 600 python bench.py > $O/bench.log 2> $O/bench.err || { tail -20 $O/bench.err; echo BENCH_FAILED; }
tail -1 $O/bench.log | cut -c1-300
