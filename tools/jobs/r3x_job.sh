set -e
out=gpurun_out/r3x
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export TW_BENCH_BACKEND=gloo
timeout -k 10 400 python3 bench.py --gpus 4 --steps 3 --warmup 1 --batch 64 --slots 32 --cpu-pairs 4 --no-extras > $out/res4.log 2>&1 || { tail -20 $out/res4.log; exit 1; }
grep '^{' $out/res4.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('resident N=4 rehearsal', d['n_gpus'], d['value'], d['config']['rehearsal'], d['roofline']['frac'], d['cpu_baseline']['value'])"
timeout -k 10 400 python3 bench.py --gpus 4 --mode queue --steps 5 --warmup 1 --batch 64 --slots 32 --cpu-pairs 4 --no-extras > $out/q4.log 2>&1 || { tail -20 $out/q4.log; exit 1; }
grep '^{' $out/q4.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('queue N=4 rehearsal', d['n_gpus'], d['value'], d['roofline']['frac'], [c['pairs'] for c in d['queue_sharded']['per_consumer']], [c['numa_node'] for c in d['queue_sharded']['per_consumer']])"
