"""What bounds S concurrent streams on one GPU: compares two rocprofv3 --kernel-trace CSVs of
`bench.py --streams S --no-extras --no-cpu-baseline --no-pcie-leg` (S = 1 and S = N) over their
steady-state part (the last 60 % of the trace: the timed chunks).

For each trace: wall time, the time with at least one kernel running, the kernel-seconds per second of
wall (average number of kernels in flight), and the same split by how much of the GPU a kernel can
fill (workgroups launched: < 256 cannot fill the 256 CUs, >= 2048 fills every CU eight times).  Per
kernel: launches, summed time, average duration in both traces and the ratio -- a kernel that
takes N times as long with N streams is serialised (it fills the GPU on its own), one that takes
the same time overlaps for free.

usage: python tools/streams_trace.py <trace_1_stream.csv> <trace_N_streams.csv> [frames_1 frames_N]"""
import csv
import re
import sys
from collections import defaultdict


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        m = re.search(r"(k_[a-z0-9_]+)", name)
        short = m.group(1) if m else None
        if short is None:
            mm = re.search(r"(radix_sort\w*|onesweep\w*|histogram\w*|fillBuffer\w*|copyBuffer\w*|elementwise\w*)", name)
            short = "lib:" + (mm.group(1) if mm else name[:24])
        wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", "256")) or 256)
        grid = int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, max(1, grid // max(wg, 1))))
    rows.sort()
    t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
    cut = t_lo + int(0.4 * (t_hi - t_lo))
    return [r for r in rows if r[0] >= cut]


def union(iv):
    iv = sorted(iv)
    if not iv:
        return 0
    busy, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return busy + ce - cs


def summarise(rows, label):
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    wall = t1 - t0
    total = sum(r[1] - r[0] for r in rows)
    print("%s: wall %.1f ms, %d launches, some kernel running %.1f %% of the wall, kernel time / wall = %.2f" % (
        label, wall / 1e6, len(rows), 100.0 * union([(r[0], r[1]) for r in rows]) / wall, total / wall))
    classes = (("< 256 workgroups (cannot fill the CUs)", 0, 256), ("256 .. 2047", 256, 2048),
               (">= 2048 (fills every CU eight times)", 2048, 1 << 60))
    for name, lo, hi in classes:
        sel = [r for r in rows if lo <= r[3] < hi]
        t = sum(r[1] - r[0] for r in sel)
        print("   %-42s %6d launches, kernel time / wall %.2f, running %.1f %% of the wall" % (
            name, len(sel), t / wall, 100.0 * union([(r[0], r[1]) for r in sel]) / wall))
    per = defaultdict(lambda: [0, 0, 0])
    for s, e, n, wgs in rows:
        p = per[n]
        p[0] += 1
        p[1] += e - s
        p[2] += wgs
    return wall, per


a = load(sys.argv[1])
b = load(sys.argv[2])
wall_a, per_a = summarise(a, "1 stream ")
wall_b, per_b = summarise(b, "N streams")
print("%-28s %8s %9s %9s | %8s %9s %9s | %s" % ("kernel", "calls", "ms", "avg us", "calls", "ms", "avg us", "avg ratio, workgroups per launch"))
for n, (c, t, w) in sorted(per_b.items(), key=lambda kv: -kv[1][1])[:30]:
    ca, ta, _ = per_a.get(n, (0, 0, 0))
    avg_a = ta / ca / 1e3 if ca else 0.0
    avg_b = t / c / 1e3
    print("%-28s %8d %9.2f %9.1f | %8d %9.2f %9.1f | %5.2f  %d" % (
        n[:28], ca, ta / 1e6, avg_a, c, t / 1e6, avg_b, (avg_b / avg_a) if avg_a else 0.0, w // c))
