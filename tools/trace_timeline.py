"""Timeline analysis of a rocprofv3 --kernel-trace CSV of tools/perf_probe.py: for the merge of the
last full chunk, how much of the wall time has no kernel running on any stream (host latency,
synchronisation bubbles), per-stream busy time, and the largest kernels / gaps."""
import csv
import re
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    m = re.search(r"(k_[a-z0-9_]+|trampoline_kernel|init_lookback_scan_state|fillBuffer\w*|copyBuffer\w*)", name)
    short = m.group(1) if m else name[:40]
    if short == "trampoline_kernel":
        mm = re.search(r"(radix_sort\w*|scan\w*|reduce_by_key\w*|unique\w*|partition\w*|lookback\w*|onesweep\w*|histogram\w*)", name)
        short = "rocprim:" + (mm.group(1) if mm else "?")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, int(r["Stream_Id"])))
rows.sort()
# merge interval of a chunk: from k_build_bucket_table to k_flatten
starts = [i for i, r in enumerate(rows) if r[2] == "k_build_bucket_table"]
ends = [i for i, r in enumerate(rows) if r[2] == "k_flatten"]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
i0 = starts[which]
i1 = [e for e in ends if e > i0][0]
seg = rows[i0:i1]
t0, t1 = seg[0][0], max(r[1] for r in seg)
print("chunk %d merge: %.2f ms, %d launches" % (which, (t1 - t0) / 1e6, len(seg)))
# union busy
ev = sorted((r[0], r[1]) for r in seg)
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
gaps = []
for s, e in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("GPU busy (any stream) %.2f ms, idle %.2f ms in %d gaps" % (busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps)))
hist = defaultdict(lambda: [0, 0])
for g, _ in gaps:
    b = 5 if g < 5000 else 10 if g < 10000 else 20 if g < 20000 else 50 if g < 50000 else 100 if g < 100000 else 1000
    hist[b][0] += 1
    hist[b][1] += g
for b in sorted(hist):
    print("  gaps <%4d us: %5d, %.2f ms" % (b, hist[b][0], hist[b][1] / 1e6))
per_stream = defaultdict(int)
per_kernel = defaultdict(lambda: [0, 0])
for s, e, n, st in seg:
    per_stream[st] += e - s
    per_kernel[n][0] += 1
    per_kernel[n][1] += e - s
print("per stream busy (ms):", {k: round(v / 1e6, 2) for k, v in per_stream.items()})
print("top kernels:")
for n, (c, t) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:28]:
    print("  %-34s %5d  %8.2f ms  avg %7.1f us" % (n, c, t / 1e6, t / c / 1e3))
# what precedes the largest gaps
print("largest gaps (us) and the kernel that ended before them:")
gl = sorted(gaps, reverse=True)[:12]
for g, at in gl:
    prev = max((r for r in seg if r[1] <= at), key=lambda r: r[1])
    nxt = min((r for r in seg if r[0] >= at + g), key=lambda r: r[0])
    print("  %7.1f after %-28s before %s" % (g / 1e3, prev[2], nxt[2]))
print("per stream x kernel (ms):")
psk = defaultdict(lambda: defaultdict(int))
for s, e, n, st in seg:
    psk[st][n] += e - s
for st in sorted(psk):
    items = sorted(psk[st].items(), key=lambda kv: -kv[1])[:14]
    print("  stream", st, ", ".join("%s %.1f" % (n, t / 1e6) for n, t in items))
# stage walk: time between consecutive k_filter launches
fl = [r for r in seg if r[2] == "k_filter"]
print("stages (k_filter to k_filter), ms:")
for a, b in zip(fl, fl[1:] + [(t1, t1, "", 0)]):
    inside = [r for r in seg if a[0] <= r[0] < b[0]]
    names = defaultdict(int)
    for r in inside:
        names[r[2]] += r[1] - r[0]
    top = sorted(names.items(), key=lambda kv: -kv[1])[:5]
    print("  %7.2f  grid %9d  %s" % ((b[0] - a[0]) / 1e6, 0, ", ".join("%s %.2f" % (n, t / 1e6) for n, t in top)))

# optional: the launches of one stage, in start order (argv[3] = stage index)
if len(sys.argv) > 3:
    st = int(sys.argv[3])
    fidx = [i for i, r in enumerate(seg) if r[2] == "k_filter"]
    lo = fidx[st]
    hi = fidx[st + 1] if st + 1 < len(fidx) else len(seg)
    base = seg[lo][0]
    print("launches of stage %d:" % st)
    for r in seg[lo:hi]:
        print("  %9.1f us  +%8.1f us  stream %d  %s" % ((r[0] - base) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[2]))
