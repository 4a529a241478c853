set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5r
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_worker_modes.py -q -m gpu --maxfail=5 -p no:cacheprovider -k "decomposition" > $OUT/tests_quick.log 2>&1
tail -2 $OUT/tests_quick.log
bash tools/measure_round.sh r5_f > gpurun_out/r5_f.log 2>&1
tail -1 gpurun_out/r5_f.log | cut -c1-160
