#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid): calls, average / min duration, share.
usage: rocprof_summary.py <dir-with-*_kernel_trace.csv> [title]   -> markdown on stdout"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else d
    files = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            wx, wy, wz = int(r["Workgroup_Size_X"]), int(r["Workgroup_Size_Y"]), int(r["Workgroup_Size_Z"])
            blocks = (int(r["Grid_Size_X"]) // wx, int(r["Grid_Size_Y"]) // wy, int(r["Grid_Size_Z"]) // wz)
            k = (r["Kernel_Name"].replace("void ", "").replace("twk::", ""), blocks, r["VGPR_Count"], r["LDS_Block_Size"])
            agg[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in agg.values())
    print("# %s\n" % title)
    print("| kernel | workgroups (x,y,z) | VGPR | LDS B | calls | avg us | min us | total ms | share |")
    print("|---|---|---|---|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("| `%s` | %s | %s | %s | %d | %.1f | %.1f | %.2f | %.1f%% |" %
              (k[0][:60], "x".join(map(str, k[1])), k[2], k[3], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3,
               sum(v) / 1e6, 100.0 * sum(v) / tot))


if __name__ == "__main__":
    main()
