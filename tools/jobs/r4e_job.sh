set -e
out=gpurun_out/r4e
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r4e/bench.json'))
print('value', d['value'], d['config']['mode'], 'ms/step', d['ms_per_step'])
print('resident_hbm', d.get('resident_hbm',{}).get('pairs_per_s'), 'sens', d.get('input_sensitivity'))
print('cfg3', d['config3_host_pinned']['pairs_per_s'], d['config3_host_pinned']['steady_state_pairs_per_s'], 'roof', d['roofline']['frac'], d['roofline_polyexp']['frac'])
PY
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --mode resident --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $out/c_$name.json 2> $out/c_$name.err || { echo "$name failed"; tail -5 $out/c_$name.err; return 0; }
  python3 -c "import json; d=json.load(open('$out/c_$name.json')); print('$name', d['value'])" | tee -a $out/chunks.txt
}
run base A=1
run l1_4 TW_CHUNK_PAIRS_LEVELS=0,4
run l1_6 TW_CHUNK_PAIRS_LEVELS=0,6
run l1_8 TW_CHUNK_PAIRS_LEVELS=0,8
run l1_16 TW_CHUNK_PAIRS_LEVELS=0,16
run l1_32 TW_CHUNK_PAIRS_LEVELS=0,32
run l2_16 TW_CHUNK_PAIRS_LEVELS=0,0,16
run l2_32 TW_CHUNK_PAIRS_LEVELS=0,0,32
run l3_64 TW_CHUNK_PAIRS_LEVELS=0,0,0,64
run l0_1 TW_CHUNK_PAIRS_LEVELS=1
run l0_1_lanes2 TW_CHUNK_PAIRS_LEVELS=1 TW_LANES=2
run l0_2_lanes2 TW_CHUNK_PAIRS_LEVELS=2 TW_LANES=2
run base2 A=1
