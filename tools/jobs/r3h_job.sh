set -e
out=gpurun_out/r3h
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in 8 16 32; do
  timeout -k 10 300 python3 tools/e2e_files.py 640 $t > $out/e2e_$t.txt 2>&1 || true
  tail -1 $out/e2e_$t.txt
done
export KB_W=3840 KB_H=2160 KB_PARAMS="winSize=50,pyrLevels=5,pyrIterations=5" KB_SLOTS=32
timeout -k 10 200 python3 tools/kbench.py 6 3 0 > $out/kbench_cfg5.txt 2>&1 || true
cat $out/kbench_cfg5.txt
tools/sq_probe.sh $out/sq_cfg5_fused 6 3 0 0 > $out/sq_cfg5_fused.txt 2>&1 || true
tools/sq_probe.sh $out/sq_cfg5_last 6 3 0 2 > $out/sq_cfg5_last.txt 2>&1 || true
tail -40 $out/sq_cfg5_fused.txt
