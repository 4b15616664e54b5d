#!/bin/bash
# Everything profiles/ is built from, in one GPU call:  tools/final_profile.sh <outdir under gpurun_out> [tag]
#   1. the default bench line (CPU baseline and the config-3 / config-5 / queue-sharded extras included)
#   2. rocprofv3 --kernel-trace --stats of the same bench (no CPU leg, no extras)
#   3. HBM traffic: two --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel trace only
#   4. SQ / GRBM counter passes of the level-0 polyexp and blur launches (tools/sq_probe.sh over tools/kbench.py)
#   5. shader clock + board power sampled every 50 ms while the bench runs (tools/clock_watch.py)
#   6. single-pair latency (tools/latency.py)
# PMC passes never combine with sys/hip/hsa tracing (kernel trace only).
out=${1:-gpurun_out/final}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 bench.py > "$out/bench.log" 2>&1; grep '^{' "$out/bench.log" | tail -1 > "$out/bench.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$out/trace.log" 2>&1
python3 tools/rocprof_summary.py "$out/trace" "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras (default mode: 256-pair steps from page-locked host memory, uploads on the copy stream; single compute stream, 2 engine batches of 128 per step, one launch per kernel and level for a whole batch)" > "$out/kernel_trace_summary.md"
cp "$out"/trace/*kernel_stats.csv "$out/kernel_stats.csv" 2>/dev/null
grep '^{' "$out/trace.log" | tail -1 > "$out/bench_under_rocprof.json"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -o run -- python3 bench.py --steps 1 --warmup 1 --batch 128 --no-cpu-baseline --no-prof --no-extras > "$out/pmc_$c.log" 2>&1
done
python3 tools/pmc_traffic.py "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE" "$out/traffic.json" "$out/pmc_hbm_traffic.md"
tools/sq_probe.sh "$out/sq_polyexp" 8 1 0 > "$out/sq_polyexp.txt" 2>&1
tools/sq_probe.sh "$out/sq_flow_iter" 8 3 0 4 > "$out/sq_flow_iter.txt" 2>&1
tools/sq_probe.sh "$out/sq_flow_iter_ups" 8 3 0 8 > "$out/sq_flow_iter_ups.txt" 2>&1
python3 tools/sq_report.py "$out/sq_polyexp" "tw_polyexp_pk<7, 8, 0>" 265420800 "tw_polyexp_pk<7,8> @ level 0, 64 pairs (128 images of 1920x1080) per launch" packed > "$out/polyexp_sq.md"
python3 tools/sq_report.py "$out/sq_flow_iter" "tw_flow_iter<15, 0" 132710400 "tw_flow_iter<15,0> (one whole iteration, no M in HBM) @ level 0, 64 pairs per launch" > "$out/flow_iter_sq.md"
python3 tools/sq_report.py "$out/sq_flow_iter_ups" "tw_flow_iter<15, 1" 132710400 "tw_flow_iter<15,1> (first iteration of a level: flow upsample fused) @ level 0, 64 pairs per launch" > "$out/flow_iter_ups_sq.md"
python3 tools/fi_stamps.py 4 > "$out/flow_iter_stamps.txt" 2>&1
# BASELINE config 5 (4K, 51-tap window): HBM traffic of its level-0 window launch
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc5_$c" -o run -- python3 tools/bench_config5.py 16 2 > "$out/pmc5_$c.log" 2>&1
done
python3 tools/pmc_kernel_traffic.py "$out/pmc5_FETCH_SIZE" "$out/pmc5_WRITE_SIZE" "tw_blur_solve4y<25" 16 "$out/traffic_cfg5.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/bench_config5.py 16 2 (tools/final_profile.sh)" > "$out/traffic_cfg5.log" 2>&1
# ... and what really holds that launch: VALU issue share + shader clock (one SQ pass; merged into traffic_cfg5.json as `_valu`)
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc5_VALU" -o run -- python3 tools/bench_config5.py 16 2 > "$out/pmc5_VALU.log" 2>&1
python3 tools/pmc_valu.py "$out/pmc5_VALU" "$out/valu_cfg5.json" "tw_blur_solve4y<25" "tw_blur_solve8<25:packed" > "$out/valu_cfg5.log" 2>&1
python3 - "$out" <<'PY'
import json, sys, os
d = sys.argv[1]
try:
    t = json.load(open(os.path.join(d, "traffic_cfg5.json"))); v = json.load(open(os.path.join(d, "valu_cfg5.json")))
    if v.get("tw_blur_solve4y<25"):
        t["_valu"] = v["tw_blur_solve4y<25"]
        json.dump(t, open(os.path.join(d, "traffic_cfg5.json"), "w"), indent=1)
except Exception as e:
    print("no _valu for config 5:", e)
PY
python3 tools/clock_watch.py "$out/clock_power.json" -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extras > "$out/clock_power.log" 2>&1
python3 tools/latency.py 40 > "$out/latency.txt" 2>&1
echo "two-stream schedule (TW_LAT_FUSED=0):" >> "$out/latency.txt"; TW_LAT_FUSED=0 python3 tools/latency.py 40 >> "$out/latency.txt" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$out/lat_trace" -o run -- python3 tools/latency.py 10 1 > "$out/lat_trace.log" 2>&1
python3 tools/timeline.py "$out/lat_trace" 23 > "$out/lat_timeline.txt" 2>&1
echo done; cut -c1-400 "$out/bench.json"
