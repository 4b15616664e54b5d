set -e
out=gpurun_out/r3u
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 __graft_entry__.py smoke > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -2 $out/smoke.log
bash tools/final_profile.sh gpurun_out/final_r03c
