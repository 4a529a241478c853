"""Summarises a rocprofv3 --pmc counter_collection.csv per kernel: launches, counter sum, per launch.

    python tools/pmc_summary.py <counter_collection.csv> [...] [--wave-json OUT]

--wave-json OUT writes the per-launch bytes of vsg::k_merge_wave to OUT and those of k_spine to OUT
with "wave" replaced by "spine", in the form bench.py reads from profiles/r2_pmc_{wave,spine}.json.

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
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            per[name][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[name].add(row["Dispatch_Id"])
    return per, launches


def main():
    out = {}
    argv = sys.argv[1:]
    wave_json = None
    if "--wave-json" in argv:
        i = argv.index("--wave-json")
        wave_json = argv[i + 1]
        del argv[i:i + 2]
    for path in argv:
        per, launches = summarise(path)
        for name, counters in per.items():
            n = len(launches[name])
            e = out.setdefault(name, {"launches": n})
            for c, v in counters.items():
                e[c + "_KB_sum"] = v
                e[c + "_KB_per_launch"] = v / max(n, 1)
    rows = sorted(out.items(), key=lambda kv: -sum(v for k, v in kv[1].items() if k.endswith("_sum")))
    print(json.dumps(dict(rows[:25]), indent=1))
    if wave_json:
        for name, e in out.items():
            for kern, tag in (("k_merge_wave", "wave"), ("k_spine", "spine")):
                if kern not in name or "k_spine_" in name:
                    continue
                json.dump({
                    "kernel": name, "launches": e["launches"],
                    "fetch_bytes_per_launch_raw": e.get("FETCH_SIZE_KB_per_launch", 0.0) * 1e3,
                    "write_bytes_per_launch_raw": e.get("WRITE_SIZE_KB_per_launch", 0.0) * 1e3,
                    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                              "'bench.py --no-cpu-baseline --no-pcie-leg', tools/measure_round.sh; "
                              "raw counters (gfx950: FETCH_SIZE may under-count wide streaming "
                              "reads by 2x, MI355X_MICROARCH.md)"},
                          open(wave_json.replace("wave", tag), "w"), indent=1)


if __name__ == "__main__":
    main()
