set -e
out=gpurun_out/r3p
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python3 tools/e2e_files.py 30000 16 > $out/soak.txt 2>&1 || true
tail -n 2 $out/soak.txt
