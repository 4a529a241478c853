#!/bin/bash
# Reproduces the round's measurement set on a GPU box:
#   tools/measure_round.sh <tag>      -> gpurun_out/<tag>/{bench.json,kernel_stats.csv,pmc_summary.json,pmc_wave.json,pmc_spine.json}
# 1. rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline --no-pcie-leg --no-extras` (the
# default workload without the CPU legs and the information-only extras), 2. FETCH_SIZE, WRITE_SIZE and
# the LDS bank-conflict counters of the same command in their own --pmc passes (they do not fit one
# pass, MI355X_MICROARCH.md), summarised per kernel -- every kernel -- by tools/pmc_summary.py together
# with the average launch durations of step 1, 3. the compact kernel table, 4. the default bench.py
# line (what the driver records) last, reading the summaries of this session.
set -u
TAG=${1:-r6}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --no-cpu-baseline --no-pcie-leg --no-extras"
rm -rf /tmp/prof_ks && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o ks -- $CMD > "$OUT/ks.log" 2>&1
cp /tmp/prof_ks/ks_kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/prof_$N
  timeout 900 rocprofv3 --pmc $C --output-format csv -d /tmp/prof_$N -o cc -- $CMD > "$OUT/pmc_$N.log" 2>&1
  cp /tmp/prof_$N/cc_counter_collection.csv "$OUT/cc_$N.csv" 2>/dev/null
done
python "$ROOT/tools/pmc_summary.py" "$OUT"/cc_*.csv --stats "$OUT/kernel_stats.csv" --wave-json "$OUT/pmc_wave.json" --bench-log "$OUT/pmc_FETCH_SIZE.log" > "$OUT/pmc_summary.json"
# 3. the compact per-kernel table bench.py embeds (4 chunk boundaries: 1 warm-up + 3 timed)
python "$ROOT/tools/kernel_table.py" "$OUT/kernel_stats.csv" "$OUT/pmc_summary.json" 4 > "$OUT/kernel_table.json"
# 4. the default bench line LAST, with this session's counter summaries in place: the traffic,
# the algorithmic bytes it is held against and the kernel table it embeds are then from one
# session on one box (bench.py reads profiles/<round>_pmc_{wave,spine}.json and
# profiles/<round>_kernel_table.json)
ROUND=${TAG%%_*}
cp "$OUT/pmc_wave.json" "$ROOT/profiles/${ROUND}_pmc_wave.json" 2>/dev/null
cp "$OUT/pmc_spine.json" "$ROOT/profiles/${ROUND}_pmc_spine.json" 2>/dev/null
cp "$OUT/kernel_table.json" "$ROOT/profiles/${ROUND}_kernel_table.json" 2>/dev/null
python "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
python "$ROOT/tools/show_stats.py" "$OUT/kernel_stats.csv" 12
rm -f "$OUT"/cc_*.csv      # several MB each; the summaries are what is kept
head -c 1500 "$OUT/bench.json"
