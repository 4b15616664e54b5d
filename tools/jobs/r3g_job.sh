set -e
out=gpurun_out/r3g
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 900 python3 bench.py > $out/bench.log 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
grep '^{' $out/bench.log | tail -1 > $out/bench.json
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r3g/bench.json"))
for k in ("value","ms_per_step"): print(k,d[k])
print(d["config"])
for k in ("roofline","roofline_polyexp"): print(k,{x:d[k][x] for x in ("frac","avg_launch_us","launches")})
for k in ("config3_host_pinned","config5_4k","queue_sharded","files_e2e","polyexp_f32_variant","scan_fused_final","cpu_baseline"):
    v=d.get(k); 
    if isinstance(v,dict): v={a:b for a,b in v.items() if a not in ("note","sample","extrapolation_note","per_consumer")}
    print(k,v)
PY
