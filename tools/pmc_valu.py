#!/usr/bin/env python3
"""VALU issue share and shader clock per kernel from ONE rocprofv3 --pmc pass (kernel trace only):

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F64 \
              SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE -- <command>
    tools/pmc_valu.py <pass dir> <out.json> [kernel-prefix[:packed] ...]

For the largest grid of every kernel whose name starts with one of the prefixes (default: the level-0 kernels of the bench):
  shader_clock_GHz  = GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md: the counter is summed over the XCDs)
  valu_issue_frac   = (f32-class wave-instructions x 2.2 + f64-class and packed-f32 ones x 4.3 cycles) / (clock cycles x 1024 SIMDs)
                      with the issue costs tools/ubench/valu_rate.hip measured on this part (docs/history.md §1); `packed`
                      after a prefix says the kernel's f32 add / mul are v_pk_* instructions (tw_polyexp_pk).
It is what profiles/r05_flow_iter_sq.md calls "VALU issue time / available SIMD time (measured issue costs)"."""
import collections
import csv
import glob
import json
import sys

DEFAULT = ["tw_flow_iter<15, 0", "tw_flow_iter<15, 1", "tw_polyexp_pk:packed", "tw_blur_solve4y", "tw_blur_solve4<", "tw_blur_solve8:packed",
           "tw_update_matrices"]


def main():
    d, outp = sys.argv[1], sys.argv[2]
    prefixes = sys.argv[3:] or DEFAULT
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].replace("void ", "").replace("twk::", ""), int(r["Grid_Size"]))
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if not dur:
        # kernel trace of the same pass: durations by (name, grid)
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = (r["Kernel_Name"].replace("void ", "").replace("twk::", ""), int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0))
                dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    out = {}
    for spec in prefixes:
        prefix, _, flag = spec.partition(":")
        cand = [k for k in acc if k[0].startswith(prefix)]
        if not cand:
            continue
        k = max(cand, key=lambda kk: (kk[1], len(acc[kk].get("SQ_INSTS_VALU", []))))
        c = {n: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for n, v in acc[k].items()}
        # one row per counter and dispatch: a dispatch's duration appears once per counter -> plain mean
        dd = dur.get(k) or [x for kk, v in dur.items() if kk[0] == k[0] for x in v]
        dd = dd[len(dd) // 4:]
        if not dd or not c.get("GRBM_GUI_ACTIVE") or not c.get("SQ_INSTS_VALU"):
            continue
        us = sum(dd) / len(dd) / 1e3
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        f64 = sum(c.get(n, 0.0) for n in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_CVT"))
        pk = (c.get("SQ_INSTS_VALU_ADD_F32", 0.0) + c.get("SQ_INSTS_VALU_MUL_F32", 0.0)) if flag == "packed" else 0.0
        f32 = c["SQ_INSTS_VALU"] - f64 - pk
        out[prefix] = {"kernel": k[0][:80], "grid_threads": k[1], "launch_us_under_pmc": round(us, 1),
                       "shader_clock_GHz": round(cyc / (us * 1e3), 3),
                       "valu_wave_instructions": round(c["SQ_INSTS_VALU"]),
                       "valu_issue_frac": round((f32 * 2.2 + (f64 + pk) * 4.3) / (cyc * 1024), 4)}
    json.dump(out, open(outp, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
