set -e
out=gpurun_out/r3c
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for g in 0 1; do
  echo "TW_LAT_GRAPH=$g" >> $out/latency.txt
  TW_LAT_GRAPH=$g timeout -k 10 120 python3 tools/latency.py 40 >> $out/latency.txt 2>&1
done
cat $out/latency.txt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/lat_trace -o run -- python3 tools/latency.py 10 1 > $out/lat_trace.log 2>&1
python3 tools/timeline.py $out/lat_trace 32 > $out/lat_timeline.txt
cat $out/lat_timeline.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
