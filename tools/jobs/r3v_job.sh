set -e
out=gpurun_out/r3v
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
FUZZ_FOCUS=win50 timeout -k 10 300 python3 tools/fuzz_parity.py 32 > $out/fuzz_win50.txt 2>&1 || true
tail -3 $out/fuzz_win50.txt
timeout -k 10 300 python3 tools/fuzz_parity.py 31 > $out/fuzz_general.txt 2>&1 || true
tail -3 $out/fuzz_general.txt
