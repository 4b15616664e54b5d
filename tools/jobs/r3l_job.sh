set -e
out=gpurun_out/r3l
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export TWFLOW_VARIANTS=1
for m in "-1,-1,-1,-1" "-1,0,-1,-1" "-1,2,-1,-1" "-1,3,-1,-1" "-1,4,-1,-1" "-1,5,-1,-1" "-1,-1,1,-1" "-1,-1,2,-1" "-1,-1,3,-1" "-1,-1,5,-1" "-1,-1,-1,1" "-1,-1,-1,2" "-1,-1,-1,3" "-1,-1,-1,5" "1,-1,-1,-1" "2,-1,-1,-1" "5,-1,-1,-1"; do
  echo "TW_BLUR_SMALL_LEVELS=$m" >> $out/sweep.txt
  TW_BLUR_SMALL_LEVELS=$m timeout -k 10 120 python3 tools/latency.py 30 1 >> $out/sweep.txt 2>&1
done
cat $out/sweep.txt
