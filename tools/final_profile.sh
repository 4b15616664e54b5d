#!/bin/bash
# Everything profiles/ is built from, in one GPU call:  tools/final_profile.sh <outdir under gpurun_out>
#   1. the default bench line (with the CPU baseline)
#   2. rocprofv3 --kernel-trace --stats of the same bench (no CPU leg)
#   3. HBM traffic: two --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel trace only
out=${1:-gpurun_out/final}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python3 bench.py > "$out/bench.log" 2>&1; grep '^{' "$out/bench.log" | tail -1 > "$out/bench.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o run -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > "$out/trace.log" 2>&1
python3 tools/rocprof_summary.py "$out/trace" "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras (single stream, engine batch 64)" > "$out/kernel_trace_summary.md"
cp "$out"/trace/*kernel_stats.csv "$out/kernel_stats.csv" 2>/dev/null
grep '^{' "$out/trace.log" | tail -1 > "$out/bench_under_rocprof.json"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-extras > "$out/pmc_$c.log" 2>&1
done
python3 tools/pmc_traffic.py "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE" "$out/traffic.json" "$out/pmc_hbm_traffic.md"
echo done; cat "$out/bench.json" | cut -c1-300
