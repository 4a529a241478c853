set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c
mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_worker_modes.py tests/test_gpu_devices.py -q -m gpu --maxfail=6 --durations=12 -p no:cacheprovider > $OUT/tests_quick.log 2>&1
echo "quick rc=$?" >> $OUT/tests_quick.log
tail -4 $OUT/tests_quick.log
bash tools/ab.sh ab/lib_r5base.so ab/lib_r5b.so 2 > $OUT/ab.log 2>&1
cat $OUT/ab.log
LIB=$ROOT/video_segment_amd/lib/libvsg_hip.so
cd /tmp && export TMPDIR=/tmp
for L in lib_r5b; do
  rm -rf /tmp/prof_$L
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$L -o ks -- python $ROOT/tools/perf_probe.py 1920 1080 79 20 > $OUT/$L.log 2>&1
  cp /tmp/prof_$L/ks_kernel_stats.csv $OUT/${L}_kernel_stats.csv 2>/dev/null
  cp /tmp/prof_$L/ks_kernel_trace.csv /tmp/${L}_trace.csv 2>/dev/null
  python $ROOT/tools/trace_timeline.py /tmp/${L}_trace.csv 2 > $OUT/${L}_timeline.txt 2>&1
done
