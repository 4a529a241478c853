#!/bin/bash
# Reproduces the round's measurement set on a GPU box:
#   tools/measure_round.sh <tag>      -> gpurun_out/<tag>/{bench.json,kernel_stats.csv,pmc_summary.json,pmc_wave.json}
# 1. default bench.py (the line the driver records), 2. rocprofv3 --kernel-trace --stats of the
# same command (CPU legs off), 3. FETCH_SIZE and WRITE_SIZE in their own --pmc passes (they do
# not fit one pass, MI355X_MICROARCH.md), summarised per kernel by tools/pmc_summary.py.
set -u
TAG=${1:-r2}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
CMD="python $ROOT/bench.py --no-cpu-baseline --no-pcie-leg --no-extras"
rm -rf /tmp/prof_ks && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o ks -- $CMD > "$OUT/ks.log" 2>&1
cp /tmp/prof_ks/ks_kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C
  timeout 900 rocprofv3 --pmc $C --output-format csv -d /tmp/prof_$C -o cc -- $CMD > "$OUT/pmc_$C.log" 2>&1
  cp /tmp/prof_$C/cc_counter_collection.csv "$OUT/cc_$C.csv" 2>/dev/null
done
python "$ROOT/tools/pmc_summary.py" "$OUT/cc_FETCH_SIZE.csv" "$OUT/cc_WRITE_SIZE.csv" --wave-json "$OUT/pmc_wave.json" > "$OUT/pmc_summary.json"
rm -f "$OUT"/cc_*.csv      # several MB each; the summaries are what is kept
cat "$OUT/bench.json"
