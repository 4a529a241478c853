"""Summarises rocprofv3 --pmc counter_collection.csv files per kernel and joins the kernel-trace
statistics of the same command:

    python tools/pmc_summary.py <counter_collection.csv> [...] [--stats kernel_stats.csv]
                                [--wave-json OUT]

Per kernel: launches, every counter (sum and per launch), and with --stats the average launch
duration and the HBM rate (FETCH_SIZE + WRITE_SIZE) / duration against the 8 TB/s peak.
--wave-json OUT writes the per-launch bytes of vsg::k_merge_wave to OUT and those of k_spine to OUT
with "wave" replaced by "spine", in the form bench.py reads from profiles/<round>_pmc_{wave,spine}.json;
--bench-log LOG: the output of the counter pass (bench.py's own JSON line), whose algorithmic bytes
per launch are stored next to the counters of that same run.

FETCH_SIZE / WRITE_SIZE are in KB (rocprofv3 derived metrics), raw: on gfx950 FETCH_SIZE counts a
wide (16 B/lane) streaming read at half its bytes (MI355X_MICROARCH.md), other widths are
uncalibrated -- the rates are lower bounds for streaming kernels."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_hash import source_hash  # noqa: E402


HBM_PEAK = 8000.0   # GB/s


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if "rocprim" in name:
        for key in ("radix_sort_onesweep", "radix_sort_block_sort", "radix_sort_merge", "scan", "reduce_by_key",
                    "partition", "lookback_scan_state", "histogram", "unique", "transform"):
            if key in name:
                return "rocprim::" + key
        return "rocprim::other"
    name = name.split("(")[0]
    if name.startswith("void "):
        name = name[5:]
    return name


def summarise(path):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = short(row["Kernel_Name"])
            per[name][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[name].add(row["Dispatch_Id"])
    return per, launches


def main():
    out = {}
    argv = sys.argv[1:]
    wave_json = stats = None
    bench_log = None
    for flag in ("--wave-json", "--stats", "--bench-log"):
        if flag in argv:
            i = argv.index(flag)
            if flag == "--wave-json":
                wave_json = argv[i + 1]
            elif flag == "--bench-log":
                bench_log = argv[i + 1]
            else:
                stats = argv[i + 1]
            del argv[i:i + 2]
    for path in argv:
        per, launches = summarise(path)
        for name, counters in per.items():
            n = len(launches[name])
            e = out.setdefault(name, {"launches": n})
            for c, v in counters.items():
                unit = "_KB" if c.endswith("_SIZE") else ""
                e[c + unit + "_sum"] = v
                e[c + unit + "_per_launch"] = v / max(n, 1)
    if stats:
        dur = collections.defaultdict(lambda: [0, 0.0])
        with open(stats, newline="") as f:
            for row in csv.DictReader(f):
                d = dur[short(row["Name"])]
                d[0] += int(row["Calls"])
                d[1] += float(row["TotalDurationNs"])
        for name, e in out.items():
            if name in dur and dur[name][0] > 0:
                avg_us = dur[name][1] / dur[name][0] / 1e3
                e["avg_launch_us"] = avg_us
                e["total_ms"] = dur[name][1] / 1e6
                kb = e.get("FETCH_SIZE_KB_per_launch", 0.0) + e.get("WRITE_SIZE_KB_per_launch", 0.0)
                if avg_us > 0 and kb > 0:
                    e["hbm_gbps_raw"] = kb * 1e3 / (avg_us * 1e-6) / 1e9
                    e["hbm_frac_of_peak_raw"] = e["hbm_gbps_raw"] / HBM_PEAK
        for e in out.values():
            if "SQ_LDS_BANK_CONFLICT_sum" in e and e.get("SQ_LDS_IDX_ACTIVE_sum", 0) > 0:
                e["lds_bank_conflict_frac"] = e["SQ_LDS_BANK_CONFLICT_sum"] / e["SQ_LDS_IDX_ACTIVE_sum"]
    rows = sorted(out.items(), key=lambda kv: -kv[1].get("total_ms", sum(
        v for k, v in kv[1].items() if k.endswith("_sum"))))
    print(json.dumps(dict(rows), indent=1))
    # The bench line the counter pass itself printed: its algorithmic bytes per launch are the ones
    # the pass's traffic is to be compared with (another run has another number of launches).
    same_run = {}
    if bench_log:
        for line in open(bench_log, errors="replace"):
            if line.startswith("{") and '"roofline"' in line:
                r = json.loads(line)["roofline"]
                tag = "spine" if "k_spine" in r["kernel"] else "wave"
                same_run[tag] = r["bytes_per_launch"]
                same_run.update(r.get("algorithmic_bytes_per_launch_by_kernel", {}))
    if wave_json:
        for name, e in out.items():
            for kern, tag in (("k_merge_wave", "wave"), ("k_spine", "spine")):
                if kern not in name or "k_spine_" in name or "_v1" in name:
                    continue
                json.dump({
                    "kernel": name, "launches": e["launches"],
                    "fetch_bytes_per_launch_raw": e.get("FETCH_SIZE_KB_per_launch", 0.0) * 1e3,
                    "write_bytes_per_launch_raw": e.get("WRITE_SIZE_KB_per_launch", 0.0) * 1e3,
                    "algorithmic_bytes_per_launch_same_run": same_run.get(tag),
                    "source_hash": source_hash(),
                    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                              "'bench.py --no-cpu-baseline --no-pcie-leg --no-extras', tools/measure_round.sh; "
                              "raw counters (gfx950: FETCH_SIZE may under-count wide streaming "
                              "reads by 2x, MI355X_MICROARCH.md)"},
                          open(wave_json.replace("wave", tag), "w"), indent=1)


if __name__ == "__main__":
    main()
