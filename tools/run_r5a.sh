set -u
mkdir -p gpurun_out/r5a
nproc > gpurun_out/r5a/nproc.txt
# 1. quick parity (the new kernels are on by default)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_worker_modes.py -q -m gpu --maxfail=6 --durations=20 -p no:cacheprovider > gpurun_out/r5a/tests_quick.log 2>&1
echo "quick rc=$?" >> gpurun_out/r5a/tests_quick.log
tail -5 gpurun_out/r5a/tests_quick.log
# 2. A/B on one box: base vs new
bash tools/ab.sh ab/lib_r5base.so ab/lib_r5a.so 2 > gpurun_out/r5a/ab.log 2>&1
cat gpurun_out/r5a/ab.log
