# round 3, first GPU pass: the GPU test suite, the bench line, and the K1 unit-load PMC passes on the shipped kernel
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; echo TESTS_FAILED; }
tail -3 $O/tests.log
timeout -k 10 500 python bench.py > $O/bench.log 2> $O/bench.err || { tail -20 $O/bench.err; echo BENCH_FAILED; }
tail -1 $O/bench.log | cut -c1-400
WW_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 timeout -k 10 300 python bench.py --no-cpu-baseline --no-streaming --sustained-s 0.5 > $O/bench_dist1.log 2> $O/bench_dist1.err || { tail -20 $O/bench_dist1.err; echo BENCH_DIST_FAILED; }
tail -1 $O/bench_dist1.log | cut -c1-300
timeout -k 10 500 bash scripts/pmc_passes.sh logmel r3a/k1_units > $O/k1_units.txt 2>&1 || echo PMC_FAILED
tail -5 $O/k1_units.txt
