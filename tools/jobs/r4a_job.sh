set -e
out=gpurun_out/r4a
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 tools/ubench/build/blur_shape > $out/blur_shape.txt 2>&1
cat $out/blur_shape.txt
