#!/bin/bash
# Usage: [SWEEP_SIZE="W H"] tools/sweep_env.sh <outfile> "VAR=a VAR2=b" "VAR=c" ...   -- one perf_probe run
# per setting (default 1920 1080; 79 frames at 1080p, 58 at larger sizes).
OUT=$1; shift
read -r SW SH <<< "${SWEEP_SIZE:-1920 1080}"
NF=79; PAT="^k=(57|76)|total"
if [ "$SW" -gt 1920 ]; then NF=58; PAT="^k=(38|57)|total"; fi
: > "$OUT"
for SETTING in "$@"; do
  echo "== $SETTING" >> "$OUT"
  env $SETTING timeout 120 python tools/perf_probe.py "$SW" "$SH" "$NF" 20 2>&1 | grep -E "$PAT" >> "$OUT"
done
cat "$OUT"
