set -e
out=gpurun_out/r3f
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export TW_LAT_GRAPH=0
for cfg in "TW_LAT_S2_LEVELS=2" "TW_LAT_S2_LEVELS=1" "TW_LAT_S2_LEVELS=0" "TW_LAT_S2_LEVELS=-1"; do
  echo "$cfg" >> $out/sweep.txt
  env $cfg timeout -k 10 120 python3 tools/latency.py 40 1 >> $out/sweep.txt 2>&1
done
cat $out/sweep.txt
TW_LAT_S2_LEVELS=0 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/lat_trace -o run -- python3 tools/latency.py 10 1 > $out/lat_trace.log 2>&1
python3 tools/timeline.py $out/lat_trace 32 > $out/lat_timeline.txt
cat $out/lat_timeline.txt
