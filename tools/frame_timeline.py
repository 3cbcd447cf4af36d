"""Timeline of one steady-state frame from a rocprofv3 kernel trace of the frame loop.
Usage: python tools/frame_timeline.py <kernel_trace.csv> [frame index from the end, default 3]
Prints every kernel between the first kernels of two consecutive frames: start / end in us
relative to the first, stream and grid size."""
import csv
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name)[:34]


def main(path, back=3):
    rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a frame starts with the selection of the tracked landmarks (a launch of its own, or the launch it shares
    # with the solve of the first hypotheses)
    marks = [i for i, r in enumerate(rows) if "select_tracked_kernel" in r["Kernel_Name"] or
             "p3p_solve_select_kernel" in r["Kernel_Name"]]
    a, b = marks[-back - 1], marks[-back]
    t0 = int(rows[a]["Start_Timestamp"])
    # include the trackers that ran just before the frame's first kernel
    lo = a
    while lo > 0 and int(rows[lo - 1]["End_Timestamp"]) > t0 - 150000:
        lo -= 1
    for r in rows[lo:b + 1]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print("%8.1f %8.1f  %6.1f us  stream %2s  grid %7d  %s" % (s, e, e - s, r["Stream_Id"],
                                                                    int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1),
                                                                    short(r["Kernel_Name"])))
    print("frame period %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
