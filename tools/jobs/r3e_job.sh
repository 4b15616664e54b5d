set -e
out=gpurun_out/r3e
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export TW_LAT_GRAPH=0
for cfg in "TW_PP_WAVES=512" "TW_PP_WAVES=1400" "TW_PP_WAVES=3000" "TW_PP_WAVES=100000" "TW_BLUR_SMALL=0" "TW_BLUR_SMALL=1" "TW_BLUR_SMALL=4"; do
  echo "$cfg" >> $out/sweep.txt
  env $cfg timeout -k 10 120 python3 tools/latency.py 30 1 >> $out/sweep.txt 2>&1
  env $cfg TW_LATENCY_STREAMS=0 timeout -k 10 120 python3 tools/latency.py 30 1 >> $out/sweep.txt 2>&1
done
cat $out/sweep.txt
