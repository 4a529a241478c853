set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5q
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 1500 python -m pytest tests -q -m gpu --maxfail=10 -p no:cacheprovider > $OUT/tests_full.log 2>&1
echo "rc=$?" >> $OUT/tests_full.log
grep -E "passed|failed|rc=" $OUT/tests_full.log | tail -3
bash tools/measure_round.sh r5_d > gpurun_out/r5_d.log 2>&1
tail -1 gpurun_out/r5_d.log | cut -c1-160
