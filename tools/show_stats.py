"""Short table of a rocprofv3 kernel_stats.csv: kernel, calls, total ms, average / max us."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("all kernels: %.1f ms in %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[:top]:
    n = r["Name"]
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", n)
    if m and "rocprim" not in n[:40]:
        short = m.group(0)
    else:
        mm = re.search(r"(radix_sort\w*|scan_impl|scan\w*|reduce_by_key\w*|unique\w*|partition\w*|lookback\w*|onesweep\w*|histogram\w*|fillBuffer\w*|copyBuffer\w*|elementwise\w*)", n)
        short = ("lib:" + mm.group(1)) if mm else n[:36]
    print("%-36s calls %6s total %9.2f ms avg %9.1f us max %9.1f us" % (
        short[:36], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, int(r["MaxNs"]) / 1e3))
