"""Condenses the output of tools/profile_round.sh (rocprofv3 CSV files) into one table per round:
per kernel the launch count, mean duration, mean FETCH_SIZE / WRITE_SIZE (KB, as rocprofv3 reports
them) and mean SQ counters.  Usage: python tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<tag>"""
import collections
import csv
import os
import re
import shutil
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+)(<[^(]*>)?\(", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    return name[:60]


def main(src, dst):
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), dst + "_kernel_stats.csv")
    table = collections.defaultdict(dict)
    for r in csv.DictReader(open(os.path.join(src, "stats", "bench_kernel_stats.csv"))):
        k = short(r["Name"])
        table[k]["calls"] = int(r["Calls"])
        table[k]["avg_us"] = float(r["AverageNs"]) / 1e3
        table[k]["pct"] = float(r["Percentage"])
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        path = os.path.join(src, sub, "bench_counter_collection.csv")
        if not os.path.exists(path):
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            for c, v in d.items():
                table[k][c] = sum(v) / len(v)
    cols = ["calls", "avg_us", "pct", "FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU",
            "SQ_INSTS_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"]
    with open(dst + "_summary.csv", "w", newline="") as f:
        w = csv.writer(f)   # (kernel names hold commas: quoted)
        w.writerow(["kernel"] + cols)
        for k, d in sorted(table.items(), key=lambda kv: -kv[1].get("pct", 0)):
            w.writerow([k] + ["%.6g" % d[c] if c in d else "" for c in cols])
    print(open(dst + "_summary.csv").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
