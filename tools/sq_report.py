#!/usr/bin/env python3
"""Markdown report of the SQ / GRBM counter passes tools/sq_probe.sh collected for ONE kernel.

    tools/sq_report.py <probe dir> <kernel-name substring> <pixels per launch> "<title>" > profiles/xxx.md

Counters are averages over the timed dispatches of that kernel (first quarter skipped as warm-up).  Units, as
MI355X_MICROARCH.md states them: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_BUSY_CYCLES is per shader engine; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections
import csv
import glob
import sys


def main():
    d, key, px, title = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
    packed = len(sys.argv) > 5 and sys.argv[5] == "packed"  # the f32 add/mul of this kernel are v_pk_* (4 cycles)
    acc = collections.defaultdict(list)
    dur = []
    names = set()
    for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                names.add(r["Kernel_Name"][:90])
    for f in glob.glob(d + "/p*/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if not acc:
        sys.exit("no dispatch of %s" % key)
    c = {k: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for k, v in acc.items()}
    dur = dur[len(dur) // 4:]
    us = sum(dur) / len(dur) / 1e3
    g = lambda k: c.get(k, 0.0)
    cyc = g("GRBM_GUI_ACTIVE") / 8.0
    ghz = cyc / (us * 1e3) if us else 0
    f64 = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_CVT")
    valu = g("SQ_INSTS_VALU")
    wc = g("SQ_WAVE_CYCLES")
    print("# %s\n" % title)
    print("Kernel: `%s`  \nAverage over the timed dispatches of `tools/sq_probe.sh` (rocprofv3 --pmc passes with --kernel-trace "
          "only); %.0f pixels per launch.\n" % (sorted(names)[0], px))
    print("| derived | value |\n|---|---|")
    print("| duration under PMC | %.1f us |" % us)
    print("| shader clock during the kernel (GRBM_GUI_ACTIVE / 8 / duration) | %.2f GHz (2.40 GHz nominal) |" % ghz)
    print("| VALU wave-instructions per pixel (SQ_INSTS_VALU x 64 / pixels) | %.1f |" % (valu * 64 / px))
    print("| ... of which f64-class (add/mul/fma f64 + converts, 4 cycles each) | %.1f |" % (f64 * 64 / px))
    print("| ... f32 add + mul (scalar 2 cycles, packed 4 cycles each) | %.1f |" %
          ((g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_FMA_F32")) * 64 / px))
    print("| LDS instructions per pixel | %.2f |" % (g("SQ_INSTS_LDS") * 64 / px))
    print("| vector memory instructions per pixel (read / write) | %.2f / %.2f |" %
          (g("SQ_INSTS_VMEM_RD") * 64 / px, g("SQ_INSTS_VMEM_WR") * 64 / px))
    if cyc:
        print("| resident waves per CU (SQ_WAVE_CYCLES x 4 / clock cycles / 256) | %.1f |" % (wc * 4 / cyc / 256))
    if wc:
        print("| share of wave time issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES) | %.0f %% |" % (100 * g("SQ_ACTIVE_INST_ANY") / wc))
        print("| share of wave time stalled at issue (SQ_WAIT_INST_ANY: dependency / pipe busy) | %.0f %% |" % (100 * g("SQ_WAIT_INST_ANY") / wc))
        print("| share of wave time parked (SQ_WAIT_ANY: s_waitcnt / barrier) | %.0f %% |" % (100 * g("SQ_WAIT_ANY") / wc))
    if g("SQ_LDS_IDX_ACTIVE"):
        print("| LDS bank-conflict cycles / LDS active cycles | %.0f %% |" % (100 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
    if cyc and valu:
        pk = (g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32")) if packed else 0.0
        f32 = valu - f64 - pk
        for label, c32, c64 in (("nominal: f32 2 cycles, f64-class and packed f32 4 cycles per wave-instruction", 2.0, 4.0),
                                ("measured issue costs 2.2 / 4.3 cycles", 2.2, 4.3)):
            busy = (f32 * c32 + (f64 + pk) * c64) / (cyc * 1024)
            print("| VALU issue time / available SIMD time (%s) | %.0f %% |" % (label, 100 * busy))
    print("\n| counter | average per launch |\n|---|---|")
    for k in sorted(c):
        print("| %s | %.0f |" % (k, c[k]))


if __name__ == "__main__":
    main()
