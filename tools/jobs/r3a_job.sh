set -e
out=gpurun_out/r3a
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 200 python3 tools/polyexp_f32.py > $out/polyexp_f32.json 2> $out/polyexp_f32.err || { tail -20 $out/polyexp_f32.err; exit 1; }
timeout -k 10 120 python3 tools/latency.py 40 > $out/latency_before.txt 2>&1
cat $out/latency_before.txt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/lat_trace -o run -- python3 tools/latency.py 10 1 > $out/lat_trace.log 2>&1
python3 tools/timeline.py $out/lat_trace 40 > $out/lat_timeline.txt
cat $out/lat_timeline.txt
