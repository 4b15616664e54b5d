set -e
out=gpurun_out/r3o
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 900 python3 bench.py > $out/bench.log 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
grep '^{' $out/bench.log | tail -1 > $out/bench.json
python3 -c "
import json; d=json.load(open('$out/bench.json'))
print(d['value'], d['config']['single_pair_latency_ms'], d['roofline']['frac'], d['roofline_polyexp']['frac'])
print({k:v for k,v in d['files_e2e'].items() if k!='note'})
"
for t in 8 16 32; do timeout -k 10 300 python3 tools/e2e_files.py 1280 $t > $out/e2e_$t.txt 2>&1 || true; tail -n 1 $out/e2e_$t.txt; done
