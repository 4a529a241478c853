set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5p
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export VSG_BENCH_EXTRAS=configs1
for i in 1 2; do
  timeout 600 python bench.py > $OUT/b_$i.json 2> $OUT/b_$i.err
  python - <<PY
import json
n=json.loads([l for l in open("$OUT/b_$i.json") if l.startswith("{")][-1])
print(n["value"], n["configs"]["configs[1]"]["ms_per_window"], n["configs"]["configs[1]"]["phase_ms_per_window"])
PY
done
