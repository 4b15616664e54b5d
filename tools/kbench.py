#!/usr/bin/env python3
"""Isolated per-kernel timings (tw_bench_stage) at 1080p: microseconds per launch and algorithmic GB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tidal-wave_amd"))
import twflow as T  # noqa: E402

# KB_W / KB_H / KB_PARAMS="winSize=50,pyrLevels=5,pyrIterations=5" / KB_SLOTS select another configuration
# (BASELINE config 5: KB_W=3840 KB_H=2160 with the parameters above)
W, H = int(os.environ.get("KB_W", 1920)), int(os.environ.get("KB_H", 1080))
PARAMS = {k: (float(v) if "." in v else int(v)) for k, v in
          (kv.split("=") for kv in os.environ.get("KB_PARAMS", "").split(",") if kv)}
SLOTS = int(os.environ.get("KB_SLOTS", 64))


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    only = int(sys.argv[2]) if len(sys.argv) > 2 else -1      # kernel class
    only_lv = int(sys.argv[3]) if len(sys.argv) > 3 else -1   # level
    only_flags = int(sys.argv[4]) if len(sys.argv) > 4 else -1
    with T.Engine(0, T.default_params(**PARAMS), slots=SLOTS) as e:
        L = e.num_levels(W, H)
        print("%-20s %5s %6s %10s %10s %8s" % ("kernel", "level", "pairs", "us/launch", "us/pair", "GB/s"))
        for kc in (T.K_PYR, T.K_POLYEXP, T.K_UPDATE_MATRICES, T.K_BLUR_SOLVE, T.K_SCAN):
            if only >= 0 and kc != only:
                continue
            for lv in range(0, L + 1):
                if kc == T.K_SCAN and lv > 0:
                    continue
                if only_lv >= 0 and lv != only_lv:
                    continue
                n = e.level_chunk(W, H, lv)
                for flags in ((0, 2, 4, 8) if kc == T.K_BLUR_SOLVE else (0,)):
                    if only_flags >= 0 and flags != only_flags:
                        continue
                    try:
                        us = e.bench_stage(kc, W, H, lv, n, iters, flags)
                    except T.TwError:
                        continue  # tw_flow_iter (flags 4 / 8) does not take this level
                    b = e.algorithmic_bytes(kc, lv, W, H) * n
                    if kc == T.K_BLUR_SOLVE:
                        w, h = W >> lv, H >> lv
                        b = (28 + (0 if flags & 2 else 52)) * w * h * n  # as built: the refreshing launch moves 80 B/px
                        if flags & 4:
                            b = 56 * w * h * n   # tw_flow_iter: flow 8 + R0 20 + R1 20 in, flow 8 out
                        if flags & 8:
                            b = 50 * w * h * n   # ... with the coarser level's flow (2 B/px) instead of this level's
                    name = T.KERNEL_NAMES[kc] + ("(no refresh)" if flags & 2 else "(M-free)" if flags & 4 else
                                                 "(M-free+ups)" if flags & 8 else "")
                    print("%-20s %5d %6d %10.1f %10.2f %8.0f" % (name[:20], lv, n, us, us / n, b / us / 1e3 if b else 0))


if __name__ == "__main__":
    main()
