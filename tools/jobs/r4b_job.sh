set -e
out=gpurun_out/r4b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 tools/ubench/build/refresh_shape > $out/refresh_shape.txt 2>&1
cat $out/refresh_shape.txt
for nt in 0 1 2 3 7; do
  if [ $nt = 0 ]; then unset TWFLOW_LIB; else export TWFLOW_LIB=$GRAFT_REPO_ROOT/tidal-wave_amd/csrc/build/libtwflow_nt$nt.so; fi
  echo "== NT=$nt" >> $out/kbench.txt
  for k in 1 2 3; do timeout -k 10 200 python3 tools/kbench.py 20 $k 0 >> $out/kbench.txt 2>&1; done
  timeout -k 10 300 python3 bench.py --mode resident --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $out/bench_nt$nt.json 2> $out/bench_nt$nt.err
  python3 -c "import json,sys; d=json.load(open('$out/bench_nt$nt.json')); print('NT=$nt value', d['value'], 'blur', d['roofline']['avg_launch_us'], 'poly', d['roofline_polyexp']['avg_launch_us'])" | tee -a $out/summary.txt
done
grep -v "^kernel" $out/kbench.txt
