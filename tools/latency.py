#!/usr/bin/env python3
"""Single-pair latency at 1080p (BASELINE config 2): device time (the Response's `time`) and wall clock of
tw_submit_dev + tw_wait for one resident pair, median of N."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
import numpy as np  # noqa: E402
import synth  # noqa: E402
import twflow as T  # noqa: E402


def main():
    if os.environ.get("TW_LAT_TORCH") == "1":  # A/B: the same loop in a process that also holds PyTorch's HIP context (bench.py's)
        import torch
        torch.cuda.init()
        torch.cuda.set_device(0)
        _t = torch.zeros(1 << 20, device="cuda")
        torch.cuda.synchronize()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    sizes = [(1920, 1080), (640, 480), (180, 117)]
    if len(sys.argv) > 2:
        sizes = sizes[:int(sys.argv[2])]
    with T.Engine(0, T.default_params(), slots=int(os.environ.get("TW_LAT_SLOTS", "1"))) as e:
        for w, h in sizes:
            a, b = synth.make_pair(0, h, w)
            da, db = e.upload(a), e.upload(b)
            dev, wall = [], []
            for _ in range(n + 3):
                t0 = time.perf_counter()
                r = e.wait(e.submit_dev(da, db, w, h, w, 10, 5.0))
                wall.append(time.perf_counter() - t0)
                dev.append(r["time"])
            print("%dx%d: device %.4f ms  wall %.4f ms  (median of %d, min device %.4f)" %
                  (w, h, np.median(dev[3:]) * 1e3, np.median(wall[3:]) * 1e3, n, min(dev[3:]) * 1e3))


if __name__ == "__main__":
    main()
