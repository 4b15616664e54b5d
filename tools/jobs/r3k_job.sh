set -e
out=gpurun_out/r3k
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export TWFLOW_VARIANTS=1
TW_BLUR_PIPE=0 tools/sq_probe.sh $out/sq_default 8 3 0 0 > $out/sq_default.txt 2>&1 || true
TW_BLUR_PIPE=109 tools/sq_probe.sh $out/sq_pipe 8 3 0 0 > $out/sq_pipe.txt 2>&1 || true
TW_BLUR_PIPE=0 PPK_ONLY=fused timeout -k 10 120 python3 tools/power_per_kernel.py > $out/power_default.md 2>&1 || true
TW_BLUR_PIPE=109 PPK_ONLY=fused timeout -k 10 120 python3 tools/power_per_kernel.py > $out/power_pipe.md 2>&1 || true
tail -3 $out/power_default.md $out/power_pipe.md
