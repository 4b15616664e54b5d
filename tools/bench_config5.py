#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: 3840x2160 pairs, pyrLevels 5, winSize 50, iters 5, resident in HBM.
Prints pairs/s and the fraction of the 8 TB/s pair roofline (6137.7 MB/pair)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
import synth  # noqa: E402
import twflow as T  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    W, H = 3840, 2160
    kw = dict(pyrLevels=5, winSize=50, pyrIterations=5)
    with T.Engine(0, T.default_params(**kw), slots=batch) as e:
        dev = []
        for i in range(2):
            a, b = synth.make_pair(i, H, W)
            dev.append((e.upload(a), e.upload(b)))

        def step():
            tickets = [e.submit_dev(dev[i % 2][0], dev[i % 2][1], W, H, W, 10, 5.0) for i in range(batch)]
            return sum(e.wait_count(t)[0] for t in tickets)

        step()
        t0 = time.perf_counter()
        flagged = 0
        for _ in range(steps):
            flagged += step()
        dt = time.perf_counter() - t0
        per_pair = e.algorithmic_bytes_pair(W, H, 10)
        v = batch * steps / dt
        print(json.dumps({"config": "3840x2160, pyrLevels 5, winSize 50, iters 5", "pairs_per_s": round(v, 2),
                          "ms_per_pair": round(1e3 / v, 3), "algorithmic_MB_per_pair": round(per_pair / 1e6, 1),
                          "frac_of_8TBps": round(v * per_pair / 8e12, 4), "levels": e.num_levels(W, H) + 1,
                          "flagged_vectors": flagged}))


if __name__ == "__main__":
    main()
