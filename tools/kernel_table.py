"""Prints a rocprofv3 *_kernel_stats.csv as a short table (kernel, calls, mean us, share)."""
import csv
import glob
import os
import sys

from summarize_profile import short


def main(path):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
    for r in list(csv.DictReader(open(path)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
        print("%-60s %7d %10.1f us %6.2f %%" % (short(r["Name"])[:60], int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                                               float(r["Percentage"])))


if __name__ == "__main__":
    main(sys.argv[1])
