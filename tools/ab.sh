#!/bin/bash
# A/B of product libraries on ONE box: tools/ab.sh <lib_a.so> <lib_b.so> [rounds]  -- alternating
# perf_probe runs (1080p, 79 frames) with each library copied over video_segment_amd/lib/libvsg_hip.so.
A=$1; B=$2; R=${3:-2}
LIB=video_segment_amd/lib/libvsg_hip.so
cp $LIB /tmp/lib_keep.so
for i in $(seq $R); do
  for L in $A $B; do
    cp $L $LIB
    echo "== $L"; timeout 200 python tools/perf_probe.py 1920 1080 79 20 2>&1 | grep -E "^k=(57|76)|total" | cut -c1-100
  done
done
cp /tmp/lib_keep.so $LIB
