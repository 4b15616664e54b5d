set -e
out=gpurun_out/r3q
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/blur_shape tools/ubench/blur_shape.hip
timeout -k 10 120 /tmp/blur_shape > $out/blur_shape.txt 2>&1
cat $out/blur_shape.txt
