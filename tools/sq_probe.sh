#!/bin/bash
# usage: tools/sq_probe.sh <outdir under gpurun_out> <kbench args...>
#   SQ / GRBM counter passes (PMC only, with --kernel-trace) over one tools/kbench.py selection; prints per-kernel
#   averages over the timed dispatches (the first quarter of the dispatches of a kernel is skipped as warm-up).
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/p$i" -o run -- python3 tools/kbench.py "$@" > "$out/p$i.log" 2>&1 || echo "pass $i failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(sys.argv[1] + "/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:60]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, v in acc.items():
    d = dur.get(k, [])
    d = d[len(d) // 4:]
    print(k, " avg duration under PMC %.1f us (n=%d)" % (sum(d) / max(len(d), 1) / 1e3, len(d)))
    for c, x in sorted(v.items()):
        x = x[len(x) // 4:]
        print("   %-32s %16.0f  (n=%d)" % (c, sum(x) / len(x), len(x)))
PY
