set -e
out=gpurun_out/r3j
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for p in 9 109; do
  TW_BLUR_PIPE=$p timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stage_blur_solve or full_flow or tile_boundary or 1080p_one or synthetic_pairs or random_medium" > $out/pytest_$p.log 2>&1 || { tail -30 $out/pytest_$p.log; exit 1; }
  tail -2 $out/pytest_$p.log
done
for p in 0 9 109 5 15 3 0; do
  echo "TW_BLUR_PIPE=$p" >> $out/kbench.txt
  TW_BLUR_PIPE=$p timeout -k 10 200 python3 tools/kbench.py 10 3 0 0 >> $out/kbench.txt 2>&1 || true
done
cat $out/kbench.txt
for p in 0 9 109; do
  echo "TW_BLUR_PIPE=$p" >> $out/bench.txt
  TW_BLUR_PIPE=$p timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])" >> $out/bench.txt
done
cat $out/bench.txt
