set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=15 > $O/tests.log 2>&1 || { grep -E "^(FAILED|ERROR)|AssertionError" $O/tests.log | head -40; echo TESTS_FAILED; }
tail -3 $O/tests.log
for S in 0 256 512 1024; do
  WW_CNN3_SUBBATCH=$S timeout -k 10 200 python scripts/bench_forward.py > $O/fwd3_sub$S.json 2> $O/fwd3_sub$S.err || { tail -5 $O/fwd3_sub$S.err; echo FWD_FAILED; }
  python - <<PY
import json
d=json.load(open("$O/fwd3_sub$S.json")); print("sub $S", round(d["clips_per_s"]), d["stages_ms"])
PY
done
