set -e
out=gpurun_out/r3d
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TW_LAT_GRAPH=0 TW_DEBUG_HOSTTIME=1 timeout -k 10 120 python3 tools/latency.py 20 1 > $out/hosttime.txt 2>&1
tail -8 $out/hosttime.txt
TW_LAT_GRAPH=0 TW_LATENCY_STREAMS=0 TW_DEBUG_HOSTTIME=1 timeout -k 10 120 python3 tools/latency.py 20 1 > $out/hosttime_1stream.txt 2>&1
tail -4 $out/hosttime_1stream.txt
