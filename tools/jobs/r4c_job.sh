set -e
out=gpurun_out/r4c
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 tools/ubench/build/refresh_shape > $out/refresh_shape.txt 2>&1
cat $out/refresh_shape.txt
