"""Summarises a rocprofv3 --pmc counter_collection.csv per kernel: launches, counter sum, per launch.

    python tools/pmc_summary.py <counter_collection.csv> [<counter_collection.csv> ...]

Counter values of FETCH_SIZE / WRITE_SIZE are in KB (rocprofv3 derived metrics)."""
import collections
import csv
import json
import sys


def summarise(path):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0]
            per[name][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[name].add(row["Dispatch_Id"])
    return per, launches


def main():
    out = {}
    for path in sys.argv[1:]:
        per, launches = summarise(path)
        for name, counters in per.items():
            n = len(launches[name])
            e = out.setdefault(name, {"launches": n})
            for c, v in counters.items():
                e[c + "_KB_sum"] = v
                e[c + "_KB_per_launch"] = v / max(n, 1)
    rows = sorted(out.items(), key=lambda kv: -sum(v for k, v in kv[1].items() if k.endswith("_sum")))
    print(json.dumps(dict(rows[:25]), indent=1))


if __name__ == "__main__":
    main()
