set -e
out=gpurun_out/r3r
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export TWFLOW_VARIANTS=1
python3 - <<'PY'
import os, sys
sys.path.insert(0, "tidal-wave_amd"); sys.path.insert(0, "oracle")
os.environ["TW_BLUR_VARIANT"] = "2"
import numpy as np, twflow as T, oracle as O, synth
kw = dict(pyrLevels=5, winSize=50, pyrIterations=5)
for (h, w) in ((540, 960), (301, 1003), (1080, 1920)):
    a, b = synth.make_pair(1, h, w)
    wx, wy = O.farneback(a, b, O.default_params(**kw))
    with T.Engine(0, T.default_params(**kw), slots=2) as e:
        gx, gy, _ = e.calculate_internal(a, b)
        tk = [e.submit(a, b, 10, 1.0) for _ in range(2)]
        got = [e.wait(t)["vector"] for t in tk]
    print(h, w, "exact" if np.array_equal(gx, wx) and np.array_equal(gy, wy) else "MISMATCH", got[0] == O.span_scan(wx, wy, 10, 1.0), got[0] == got[1])
PY
export KB_W=3840 KB_H=2160 KB_PARAMS="winSize=50,pyrLevels=5,pyrIterations=5" KB_SLOTS=32
for v in 4 2 4 2; do
  echo "TW_BLUR_VARIANT=$v" >> $out/kbench_cfg5.txt
  TW_BLUR_VARIANT=$v timeout -k 10 200 python3 tools/kbench.py 6 3 0 >> $out/kbench_cfg5.txt 2>&1 || true
done
cat $out/kbench_cfg5.txt
for v in 4 2; do TW_BLUR_VARIANT=$v timeout -k 10 300 python3 tools/bench_config5.py 16 4; done
