set -e
out=gpurun_out/r3i
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in 8 16; do
  timeout -k 10 300 python3 tools/e2e_files.py 640 $t > $out/e2e_$t.txt 2>&1 || true
  tail -n 1 $out/e2e_$t.txt
done
export KB_W=3840 KB_H=2160 KB_PARAMS="winSize=50,pyrLevels=5,pyrIterations=5" KB_SLOTS=32 TWFLOW_VARIANTS=1
for v in 4 256 255 254 258 259 4; do
  echo "TW_BLUR_VARIANT=$v" >> $out/kbench_cfg5.txt
  TW_BLUR_VARIANT=$v timeout -k 10 200 python3 tools/kbench.py 6 3 0 >> $out/kbench_cfg5.txt 2>&1 || true
done
cat $out/kbench_cfg5.txt
