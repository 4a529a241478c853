#!/bin/bash
# Usage: tools/sweep_env.sh <outfile> "VAR=a VAR2=b" "VAR=c" ...   -- one perf_probe run per setting.
OUT=$1; shift
: > "$OUT"
for SETTING in "$@"; do
  echo "== $SETTING" >> "$OUT"
  env $SETTING timeout 120 python tools/perf_probe.py 1920 1080 79 20 2>&1 | grep -E "^k=(57|76)|total" >> "$OUT"
done
cat "$OUT"
