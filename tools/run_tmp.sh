set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5n
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
bash tools/sweep_env.sh $OUT/sweep.log "VSG_WINDOWS=1" "VSG_WINDOWS=2" "VSG_WINDOWS=3" "VSG_WINDOWS=6" "VSG_WINDOWS=12" "VSG_WINDOWS=16 VSG_SPINE_MIN=8192" "VSG_WINDOWS=24 VSG_SPINE_MIN=8192" > /dev/null 2>&1
cut -c1-100 $OUT/sweep.log
VSG_DEBUG_STAGES=1 VSG_WINDOWS=2 timeout 120 python tools/perf_probe.py 1920 1080 41 20 2>&1 | grep -E "stage b=[012] " | tail -12 | cut -c1-200
