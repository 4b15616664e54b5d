#!/bin/bash
# usage: tools/pmc_probe.sh <outdir> <kbench args...>   -- a few memory-pipeline counter passes over tools/kbench.py
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_TCP_LATENCY_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/p$i" -o run -- python3 tools/kbench.py "$@" > "$out.p$i.log" 2>&1 || echo "pass $i failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c, x in sorted(v.items()):
        x = x[len(x)//4:]  # skip warm-up dispatches
        print("   %-44s %14.0f  (n=%d)" % (c, sum(x) / len(x), len(x)))
PY
