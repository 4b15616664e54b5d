set -e
out=gpurun_out/r3t
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export TWFLOW_VARIANTS=1
for v in 4 2 4 2 4 2; do
  echo -n "TW_BLUR_VARIANT=$v " >> $out/ab.txt
  TW_BLUR_VARIANT=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])" >> $out/ab.txt
done
cat $out/ab.txt
