#!/usr/bin/env python3
"""Timeline of the last K kernel dispatches of a rocprofv3 --kernel-trace CSV: start/end in microseconds relative to
the first of them, queue id, grid.  tools/timeline.py <dir or csv> [K]"""
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-k:]
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        grid = "%sx%sx%s" % (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Grid_Size_Y"], r["Grid_Size_Z"])
        print("%8.1f %8.1f %6.1f  q%-3s %-14s %s" % (s, e, e - s, r.get("Queue_Id", "?"), grid, r["Kernel_Name"][:60]))


if __name__ == "__main__":
    main()
