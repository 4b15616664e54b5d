set -e
out=gpurun_out/r4d
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 tools/ubench/build/mall_probe > $out/mall_probe.txt 2>&1
cat $out/mall_probe.txt
