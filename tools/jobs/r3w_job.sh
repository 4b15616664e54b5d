set -e
out=gpurun_out/r3w
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python3 tools/clock_watch.py $out/clock_power_sustained.json -- python3 bench.py --steps 3000 --warmup 2 --no-cpu-baseline --no-extras > $out/sustained.log 2>&1
grep '^{' $out/sustained.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sustained', d['value'], d['steps'], d['ms_per_step'], d['roofline']['frac'], d['roofline_polyexp']['frac'])"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r3w/clock_power_sustained.json"))
s=[x for x in d["samples"] if float(x.get("power1_input",0))>1.0e9]
import statistics
print("loaded samples", len(s), "median W", statistics.median(float(x["power1_input"])/1e6 for x in s), "median MHz", statistics.median(float(x["freq1_input"])/1e6 for x in s))
n=len(s); 
for a,b in ((0,n//10),(n//2-n//20,n//2+n//20),(n-n//10,n)):
    seg=s[a:b]; print("segment", a, b, "MHz", statistics.median(float(x["freq1_input"])/1e6 for x in seg), "W", statistics.median(float(x["power1_input"])/1e6 for x in seg))
PY
