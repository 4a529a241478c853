set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_presmooth_gaussian.py -q -m gpu --maxfail=5 -p no:cacheprovider > $OUT/tests_quick.log 2>&1
tail -3 $OUT/tests_quick.log
bash tools/ab.sh ab/lib_r5d.so ab/lib_r5e.so 2 > $OUT/ab.log 2>&1
cat $OUT/ab.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_e
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o ks -- python $GRAFT_REPO_ROOT/tools/perf_probe.py 1920 1080 41 20 > $OUT/prof.log 2>&1
grep -E "k_bilateral|k_minmax" /tmp/prof_e/ks_kernel_stats.csv | cut -c1-200
