set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5b
mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
LIB=$ROOT/video_segment_amd/lib/libvsg_hip.so
cp $LIB /tmp/lib_keep.so
cd /tmp && export TMPDIR=/tmp
for L in lib_r5base lib_r5a; do
  cp $ROOT/ab/$L.so $LIB
  rm -rf /tmp/prof_$L
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$L -o ks -- python $ROOT/tools/perf_probe.py 1920 1080 79 20 > $OUT/$L.log 2>&1
  cp /tmp/prof_$L/ks_kernel_stats.csv $OUT/${L}_kernel_stats.csv 2>/dev/null
  grep -E "^k=|total" $OUT/$L.log | cut -c1-110
done
cp /tmp/lib_keep.so $LIB
