"""Top kernels of a profiled bench run as the compact table bench.py embeds in its JSON line
(roofline.kernels):

    python tools/kernel_table.py <kernel_stats.csv> <pmc_summary.json> <chunks> [top]

chunks = chunk boundaries of the profiled command (warm-up + timed steps); per kernel: launches per
step, ms per step, raw HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, separate --pmc passes of the
same command), GB/s and the fraction of the 8 TB/s peak."""
import csv
import json
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_summary import short  # noqa: E402
from source_hash import source_hash  # noqa: E402

HBM_PEAK = 8000.0


def main():
    stats, pmc_path, chunks = sys.argv[1], sys.argv[2], float(sys.argv[3])
    top = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    pmc = json.load(open(pmc_path))
    per = {}
    for row in csv.DictReader(open(stats, newline="")):
        name = short(row["Name"])
        if name.startswith("at::") or "elementwise" in name or name.startswith("(anonymous"):
            continue   # the synthetic frame generator of the bench (torch), outside the timed region
        e = per.setdefault(name, [0, 0.0])
        e[0] += int(row["Calls"])
        e[1] += float(row["TotalDurationNs"])
    rows = []
    for name, (calls, ns) in sorted(per.items(), key=lambda kv: -kv[1][1])[:top]:
        p = pmc.get(name, {})
        kb = p.get("FETCH_SIZE_KB_per_launch", 0.0) + p.get("WRITE_SIZE_KB_per_launch", 0.0)
        avg_us = ns / calls / 1e3
        gbps = kb * 1e3 / (avg_us * 1e-6) / 1e9 if kb > 0 else None
        rows.append({"kernel": name, "launches_per_step": calls / chunks, "ms_per_step": ns / 1e6 / chunks,
                     "avg_launch_us": avg_us, "hbm_bytes_per_launch_raw": kb * 1e3 if kb > 0 else None,
                     "hbm_gbps_raw": gbps, "frac_of_peak": gbps / HBM_PEAK if gbps else None,
                     "lds_bank_conflict_frac": p.get("lds_bank_conflict_frac")})
    total_ms = sum(v[1] for v in per.values()) / 1e6 / chunks
    print(json.dumps({"source": "rocprofv3 --kernel-trace --stats and --pmc FETCH_SIZE / WRITE_SIZE / LDS "
                                "bank conflicts (separate passes) of 'bench.py --no-cpu-baseline --no-pcie-leg "
                                "--no-extras', tools/measure_round.sh; raw counters",
                      "source_hash": source_hash(), "chunks_profiled": chunks, "all_kernels_ms_per_step": total_ms,
                      "launches_per_step": sum(v[0] for v in per.values()) / chunks, "top": rows}, indent=1))


if __name__ == "__main__":
    main()
