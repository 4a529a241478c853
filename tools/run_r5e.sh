set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5e
mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_worker_modes.py -q -m gpu --maxfail=6 -p no:cacheprovider -k "decomposition" > $OUT/tests_quick.log 2>&1
echo "quick rc=$?" >> $OUT/tests_quick.log
tail -3 $OUT/tests_quick.log
timeout 900 python -m pytest tests/test_gpu_configs.py -q -m gpu --maxfail=6 -p no:cacheprovider --durations=8 -k "bench_workload_1080p or bench_reported or 2560" > $OUT/tests_full.log 2>&1
echo "full rc=$?" >> $OUT/tests_full.log
tail -12 $OUT/tests_full.log
bash tools/ab.sh ab/lib_r5base.so ab/lib_r5c.so 2 > $OUT/ab.log 2>&1
cat $OUT/ab.log
cd /tmp && export TMPDIR=/tmp
for L in lib_r5c; do
  rm -rf /tmp/prof_$L
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$L -o ks -- python $ROOT/tools/perf_probe.py 1920 1080 79 20 > $OUT/$L.log 2>&1
  cp /tmp/prof_$L/ks_kernel_stats.csv $OUT/${L}_kernel_stats.csv 2>/dev/null
  cp /tmp/prof_$L/ks_kernel_trace.csv /tmp/${L}_trace.csv 2>/dev/null
  python $ROOT/tools/trace_timeline.py /tmp/${L}_trace.csv 2 > $OUT/${L}_timeline.txt 2>&1
done
