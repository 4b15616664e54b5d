set -e
out=gpurun_out/r3n
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/store_rate tools/ubench/store_rate.hip
timeout -k 10 120 /tmp/store_rate > $out/store_rate.txt 2>&1
cat $out/store_rate.txt
timeout -k 10 120 /tmp/store_rate >> $out/store_rate.txt 2>&1
