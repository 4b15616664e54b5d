set -e
out=gpurun_out/r3b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cus in 0 16 24 28; do
  echo "TW_LAT_CUS=$cus" >> $out/latency.txt
  TW_LAT_CUS=$cus timeout -k 10 120 python3 tools/latency.py 40 1 >> $out/latency.txt 2>&1
done
cat $out/latency.txt
TW_LAT_CUS=24 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/lat_trace -o run -- python3 tools/latency.py 10 1 > $out/lat_trace.log 2>&1
python3 tools/timeline.py $out/lat_trace 32 > $out/lat_timeline.txt
cat $out/lat_timeline.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
